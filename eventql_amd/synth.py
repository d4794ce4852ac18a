"""Synthetic benchmark table of SURVEY.md section 8c(ii) / 8d (numpy, host side).

  x_0 = seed; per row: x ^= x << 13; x ^= x >> 7; x ^= x << 17  (xorshift64)
  k = x % 1000, a = (x >> 8) & 0xffff, b = (x >> 24) & 0xffff,
  v = (x >> 40) / 1024.0, n = i (NULL when i % 3 == 0), s = "g" + str(k)

The device generator (evql_table_generate) produces the same sequence with a
GF(2) jump-ahead; this module is the host twin used to build small tables and
to check the device generator.
"""
import numpy as np

SEED = 88172645463325252
MASK = (1 << 64) - 1


def xorshift64_sequence(n, seed=SEED):
    """x_1 .. x_n (state after each step)"""
    out = np.empty(n, dtype=np.uint64)
    x = seed
    for i in range(n):
        x ^= (x << 13) & MASK
        x ^= x >> 7
        x ^= (x << 17) & MASK
        out[i] = x
    return out


def _step_matrix():
    """64x64 GF(2) matrix of one xorshift step as 64 column images"""
    cols = []
    for bit in range(64):
        x = 1 << bit
        x ^= (x << 13) & MASK
        x ^= x >> 7
        x ^= (x << 17) & MASK
        cols.append(x)
    return cols


def _mat_apply(cols, x):
    r = 0
    b = 0
    while x:
        if x & 1:
            r ^= cols[b]
        x >>= 1
        b += 1
    return r


def _mat_mul(a, b):
    """(a o b): apply b first, then a"""
    return [_mat_apply(a, c) for c in b]


def jump_matrices(nbits=40):
    """M^(2^j) for j < nbits, each as 64 column images (python ints)"""
    m = _step_matrix()
    out = [m]
    for _ in range(1, nbits):
        m = _mat_mul(m, m)
        out.append(m)
    return out


def xorshift64_sequence_fast(n, seed=SEED, chunk=4096):
    """vectorised: jump to the start of every chunk, then step all chunks in
    lock-step with numpy"""
    nchunks = (n + chunk - 1) // chunk
    # state at the start of chunk c = M^(c*chunk) seed
    mats = jump_matrices(48)
    starts = np.empty(nchunks, dtype=np.uint64)
    # incremental: M^chunk applied repeatedly
    mc = None
    e, j = chunk, 0
    while e:
        if e & 1:
            mc = mats[j] if mc is None else _mat_mul(mats[j], mc)
        e >>= 1
        j += 1
    x = seed
    for c in range(nchunks):
        starts[c] = x
        x = _mat_apply(mc, x)
    out = np.empty((nchunks, chunk), dtype=np.uint64)
    s = starts.copy()
    for i in range(chunk):
        s ^= s << np.uint64(13)
        s ^= s >> np.uint64(7)
        s ^= s << np.uint64(17)
        out[:, i] = s
    return out.reshape(-1)[:n]


def items_table_image(nrec, seed=3):
    """BASELINE.json configs[4]: records with a REPEATED RECORD
    items{position, price} (rlevel_max 1, dlevel_max 2), geometric 0..8 items per
    record, plus a required top-level id and an optional score.  Returns
    (cstable image bytes, stats dict).  Written by the host writer."""
    import eventql_amd as E
    from eventql_amd import capi as K
    rng = np.random.default_rng(seed)
    cnt = np.minimum(rng.geometric(0.35, nrec) - 1, 8)
    # a record without items still has one (r=0, d=0) slot
    slots = np.maximum(cnt, 1)
    total = int(slots.sum())
    starts = np.concatenate([[0], np.cumsum(slots)[:-1]])
    rl = np.ones(total, np.uint64)
    rl[starts] = 0
    rec_of_slot = np.repeat(np.arange(nrec), slots)
    dl = np.where(cnt[rec_of_slot] > 0, 2, 0).astype(np.uint64)
    pos = (np.arange(total) - starts[rec_of_slot] + 1).astype(np.uint64)
    price = rng.integers(1, 100000, total).astype(np.uint64)
    ids = np.arange(nrec, dtype=np.uint64) * np.uint64(7)
    score = rng.random(nrec) * 100.0
    w = E.Writer([
        dict(name="id", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_PLAIN),
        dict(name="items.position", logical_type=K.COL_UNSIGNED_INT,
             storage_type=K.ENC_UINT32_BITPACKED, rlevel_max=1, dlevel_max=2,
             bitpack_max_value=15),
        dict(name="items.price", logical_type=K.COL_UNSIGNED_INT,
             storage_type=K.ENC_UINT64_LEB128, rlevel_max=1, dlevel_max=2),
        dict(name="score", logical_type=K.COL_FLOAT, storage_type=K.ENC_FLOAT_IEEE754,
             dlevel_max=1)])
    w.put("id", ids)
    w.put("items.position", pos, rlvl=rl, dlvl=dl)
    w.put("items.price", price, rlvl=rl, dlvl=dl)
    w.put("score", score, present=(np.arange(nrec) % 9 != 0).astype(np.uint8))
    w.commit(nrec)
    img = w.image()
    w.close()
    defined = dl == 2
    return img, dict(cnt=cnt, total=total, n_items=int(defined.sum()),
                     sum_price=int(price[defined].sum()), sum_pos=int(pos[defined].sum()))


def table_columns(n, seed=SEED, k_mod=1000):
    x = xorshift64_sequence_fast(n, seed)
    k = x % np.uint64(k_mod)
    a = (x >> np.uint64(8)) & np.uint64(0xFFFF)
    b = (x >> np.uint64(24)) & np.uint64(0xFFFF)
    v = (x >> np.uint64(40)).astype(np.float64) / 1024.0
    return dict(x=x, k=k, a=a, b=b, v=v)


def device_string_keys(u, prefix=b"g"):
    """strings prefix + decimal(u[i]) for a torch int64 tensor `u` on the GPU, in the
    form evql_table_from_device_columns takes for a STRING_PLAIN column: (words, heap)
    with words[i] = (len << 40) | offset into the uint8 heap.  Input generation for the
    benchmark / tests (torch is plumbing here); SURVEY.md 8c(ii): s = "g" + k."""
    import torch
    n = u.numel()
    p = len(prefix)
    ndig = torch.ones_like(u)
    lim = 10
    for _ in range(18):
        ndig += (u >= lim).to(torch.int64)
        lim *= 10
    lens = ndig + p
    offs = torch.cumsum(lens, 0) - lens
    total = int((offs[-1] + lens[-1]).item()) if n else 0
    heap = torch.zeros(max(total, 1), dtype=torch.uint8, device=u.device)
    for j, ch in enumerate(prefix):
        heap[offs + j] = ch
    maxd = int(ndig.max().item()) if n else 0
    for d in range(maxd):
        # digit d counted from the most significant one, for the rows that have it
        sel = ndig > d
        us, nd, of = u[sel], ndig[sel], offs[sel]
        digit = torch.div(us, torch.pow(10, nd - 1 - d), rounding_mode="floor") % 10
        heap[of + p + d] = (digit + 48).to(torch.uint8)
    words = (lens << 40) | offs
    return words, heap
