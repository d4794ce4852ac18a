"""LSM row filters (PartitionCursor::openNextTable, partition_cursor.cc:160-195):
the device build (evql_lsm_chain_*) against the oracle's sequential
restatement, and the filtered scans merged across the chain."""
import hashlib

import numpy as np
import pytest

import eventql_amd as E
from eventql_amd import capi as K
from eventql_amd.plan import Plan, col, count, sum_, max_
import oracle_lib as O
import tables as T

S = dict(k=K.T_UINT64, a=K.T_UINT64)


def lsm_table(ids, is_update, skip, k, a, with_skip_column=True, nullable_flags=False):
    """one LSM file: __lsm_id (20 raw bytes), __lsm_is_update, __lsm_skip + payload"""
    n = len(ids)
    specs = [dict(name="__lsm_id", logical_type=K.COL_STRING, storage_type=K.ENC_STRING_PLAIN),
             dict(name="__lsm_is_update", logical_type=K.COL_BOOLEAN,
                  storage_type=K.ENC_BOOLEAN_BITPACKED, dlevel_max=1 if nullable_flags else 0),
             dict(name="k", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_PLAIN),
             dict(name="a", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_LEB128)]
    if with_skip_column:
        specs.append(dict(name="__lsm_skip", logical_type=K.COL_BOOLEAN,
                          storage_type=K.ENC_BOOLEAN_BITPACKED))
    w = E.Writer(specs)
    w.put("__lsm_id", list(ids))
    if nullable_flags:
        w.put("__lsm_is_update", np.asarray(is_update, np.uint64), present=np.ones(n, np.uint8))
    else:
        w.put("__lsm_is_update", np.asarray(is_update, np.uint64))
    w.put("k", np.asarray(k, np.uint64))
    w.put("a", np.asarray(a, np.uint64))
    if with_skip_column:
        w.put("__lsm_skip", np.asarray(skip, np.uint64))
    w.commit(n)
    img = w.image()
    w.close()
    return img


def sha(i):
    return hashlib.sha1(b"row%d" % i).digest()


def model(chain):
    """the reference loop, in python: chain = [(ids, upd, skip)] newest first"""
    seen, out = set(), []
    for ids, upd, skip in chain:
        f = np.zeros(len(ids), bool)
        for i, (d, u, s) in enumerate(zip(ids, upd, skip)):
            if s or d in seen:
                continue
            if u:
                seen.add(d)
            f[i] = True
        out.append(f)
    return out


def make_chain(seed, sizes, n_ids):
    rng = np.random.default_rng(seed)
    chain = []
    for n in sizes:
        who = rng.integers(0, n_ids, n)
        ids = [sha(int(i)) for i in who]
        upd = rng.random(n) < 0.4
        skip = rng.random(n) < 0.1
        k = who % 17
        a = rng.integers(0, 1 << 40, n)
        chain.append((ids, upd, skip, k, a))
    return chain


def test_oracle_restatement_against_the_python_model(built):
    # hand-made: X updated in the newest file shadows the older X rows; a skipped
    # update does not shadow; a non-update never shadows
    X, Y, Z = sha(1), sha(2), sha(3)
    newest = ([X, Y, X, Z], [1, 0, 1, 1], [0, 0, 0, 1])
    older = ([X, Y, Z, Z, Y], [0, 1, 1, 0, 0], [0, 0, 0, 0, 0])
    imgs = [lsm_table(ids, u, s, range(len(ids)), range(len(ids))) for ids, u, s in (newest, older)]
    got = O.oracle_lsm_filters(imgs, [True, True])
    assert got[0].tolist() == [True, True, False, False]      # 2nd X shadowed in-file, Z skipped
    assert got[1].tolist() == [False, True, True, False, False]
    chain = make_chain(3, [5000, 3000, 8000], 4000)
    imgs = [lsm_table(*c) for c in chain]
    exp = model([(c[0], c[1], c[2]) for c in chain])
    got = O.oracle_lsm_filters(imgs, [True] * 3)
    for g, e in zip(got, exp):
        assert (g == e).all()
    # ids must be 20 bytes (SHA1Hash ctor raises)
    bad = lsm_table([b"short"], [0], [0], [1], [1])
    with pytest.raises(RuntimeError):
        O.oracle_lsm_filters([bad], [True])


@pytest.mark.gpu
def test_device_filters_match_the_oracle(ctx):
    chain = make_chain(11, [3, 70_001, 30_000, 150_000], 60_000)
    # table 0 plays an arena (skiplist from memory, no skip column; arenas come first,
    # partition_cursor.cc:91-132); table 2 has no skiplist at all; nullable flag columns
    # go through the decode-to-SoA path
    arena_skip = np.array([0, 1, 0], np.uint8)
    imgs = [lsm_table(*chain[0], with_skip_column=False),
            lsm_table(*chain[1]),
            lsm_table(*chain[2], with_skip_column=False, nullable_flags=True),
            lsm_table(*chain[3])]
    has_skip = [False, True, False, True]
    skips = [arena_skip, None, None, None]
    exp = O.oracle_lsm_filters(imgs, has_skip, skips)
    tables = [ctx.open_image(i) for i in imgs]
    ch = E.LsmChain(ctx)
    for t, h, sk in zip(tables, has_skip, skips):
        ch.add(t, h, sk)
    ch.build()
    total_kept = 0
    for i, e in enumerate(exp):
        f, kept = ch.filter(i)
        assert (f == e).all(), i
        assert kept == int(e.sum())
        total_kept += kept
    assert 0 < total_kept < sum(len(c[0]) for c in chain)

    # the filtered scans, merged across the chain on the device, against the oracle
    kw = dict(select=[col("k"), count(1), sum_(col("a")), max_(col("a"))], group_by=[col("k")])
    acc = {}
    for img, e in zip(imgs, exp):
        for k, c, s, m in O.oracle_run(img, Plan(S, row_filter=e, **kw)).rows():
            c0, s0, m0 = acc.get(k, (0, 0, 0))
            acc[k] = (c0 + c, (s0 + s) & 0xFFFFFFFFFFFFFFFF, max(m0, m))
    import torch
    queries = [t.query(Plan(S, row_filter=ch.filter(i)[0], **kw)) for i, t in enumerate(tables)]
    for q in queries:
        q.execute()
    rw = queries[0].record_words()
    buf = torch.zeros(64 * rw, dtype=torch.int64, device="cuda")
    for q in queries[1:]:
        n = q.export_groups(buf.data_ptr(), 64)
        if n:
            queries[0].import_groups(buf.data_ptr(), n)
    got = {r[0]: r[1:] for r in queries[0].fetch_all().rows()}
    assert got == acc
    for q in queries:
        q.close()

    # ids of the wrong length are an error, as in the reference
    bad = ctx.open_image(lsm_table([b"short", sha(1)], [0, 1], [0, 0], [1, 2], [1, 2]))
    ch2 = E.LsmChain(ctx)
    ch2.add(bad, True)
    with pytest.raises(E.EvqlError) as ei:
        ch2.build()
    assert ei.value.code == K.EVQL_ERUNTIME and "invalid SHA1Hash" in ei.value.msg
    ch2.close()
    ch.close()
    bad.close()
    for t in tables:
        t.close()
