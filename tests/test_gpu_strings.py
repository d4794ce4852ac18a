"""STRING_PLAIN value boundaries found on the device (page_reader_lenencstring.cc:37-62):
chunk tables + pointer doubling, no host copy of the file.  Every case is checked against
the oracle's sequential reader through a GROUP BY on the string column (group keys =
the exact bytes of every distinct value, counts = how many rows got them) and a first-row
select (value of a given row)."""
import numpy as np
import pytest

import eventql_amd as E
from eventql_amd import capi as K
from eventql_amd.plan import Plan, col, count, sum_, lit, Call
import oracle_lib as O
import tables as T

pytestmark = pytest.mark.gpu

S = dict(s=K.T_STRING, x=K.T_UINT64)


def build(vals, present=None):
    n = len(vals)
    w = E.Writer([dict(name="s", logical_type=K.COL_STRING, storage_type=K.ENC_STRING_PLAIN,
                       dlevel_max=0 if present is None else 1),
                  dict(name="x", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_PLAIN)])
    w.put("s", vals, present=present)
    w.put("x", np.arange(n, dtype=np.uint64))
    w.commit(n)
    img = w.image()
    w.close()
    return img


def check_table(ctx, img, extra=()):
    t = ctx.open_image(img)
    try:
        s, x = col("s"), col("x")
        plans = [
            Plan(S, select=[s, count(1), sum_(x)], group_by=[s]),
            # one group per row: every single boundary matters
            Plan(S, select=[x, s, count(1)], group_by=[x], groups_hint=400000),
        ] + list(extra)
        for plan in plans:
            exp = O.oracle_run(img, plan)
            q = t.query(plan)
            try:
                got = q.run()
                assert got.nrows == exp.nrows
                T.compare_results(got.rows(), exp.rows(), exp.types, key_cols=1)
            finally:
                q.close()
    finally:
        t.close()


def test_short_strings_many_pages(ctx):
    rng = np.random.default_rng(11)
    n = 300_000
    u = rng.integers(0, 10_000_000, n)
    vals = [b"g%d" % k for k in u]
    img = build(vals)
    assert len(img) > 4 * 512 * 1024
    check_table(ctx, img, extra=[Plan(S, select=[count(1), sum_(col("x"))],
                                      where=Call("gte", col("s"), lit(b"g5")))])


def test_every_header_width_and_chunk_border(ctx):
    """lengths 0, 1, 127 (1-byte header), 128 .. 16383 (2 bytes), 16384+ (3 bytes);
    values that end exactly at, one before and one behind 4096-byte chunk borders and
    512 KiB page borders"""
    rng = np.random.default_rng(12)
    lens = [0, 1, 2, 126, 127, 128, 129, 255, 256, 1000, 1022, 1023, 1024, 1025, 4090, 4095,
            4096, 4097, 8191, 8192, 16383, 16384, 16385, 40000, 61000, 65535, 65536, 70000,
            200000, 600000]
    vals = []
    for rep in range(6):
        for L in lens:
            vals.append(bytes(rng.integers(0, 256, L, dtype=np.uint8)))
            # a run of short values behind every long one re-synchronises differently
            for _ in range(int(rng.integers(0, 40))):
                vals.append(b"q%d" % rng.integers(0, 1000))
    # fillers that put the following value at chosen offsets relative to chunk borders
    for target in (4096 * 3, 4096 * 3 - 1, 4096 * 3 + 1, 512 * 1024, 512 * 1024 - 1,
                   512 * 1024 + 1):
        vals.append(b"z" * 50)
    img = build(vals)
    check_table(ctx, img)


def test_bytes_that_look_like_headers(ctx):
    """content bytes with the continuation bit set everywhere: every byte offset decodes
    as the start of some (wrong) value; only the chain from offset 0 is right"""
    rng = np.random.default_rng(13)
    n = 120_000
    vals = [bytes(rng.integers(0x80, 0x100, int(rng.integers(0, 24)), dtype=np.uint8))
            for _ in range(n)]
    check_table(ctx, build(vals))


def test_nullable_strings(ctx):
    rng = np.random.default_rng(14)
    n = 250_000
    vals = [b"s%d" % (k % 5000) for k in rng.integers(0, 1 << 30, n)]
    present = (rng.random(n) < 0.7).astype(np.uint8)
    img = build(vals, present=present)
    check_table(ctx, img)
    # all NULL and all present
    check_table(ctx, build(vals[:5000], present=np.zeros(5000, dtype=np.uint8)))
    check_table(ctx, build(vals[:5000], present=np.ones(5000, dtype=np.uint8)))


def test_tiny_and_empty(ctx):
    check_table(ctx, build([b"only"]))
    check_table(ctx, build([b"", b"", b""]))
    check_table(ctx, build([b"x" * 5000]))  # one value longer than a chunk


def test_truncated_string_stream_is_an_error(ctx):
    """fewer bytes than the lengths promise: the reference raises 'end of column reached'"""
    vals = [b"abcdefgh" * 8 for _ in range(20_000)]
    img = bytearray(build(vals))
    # flip a header in the last data page to a huge length
    n, _ = E.inspect_image(bytes(img))
    t = ctx.open_image(bytes(img))
    info = [c for c in t.columns() if c["name"] == "s"][0]
    t.close()
    assert info["n_data_pages"] >= 2
    # find the first value header: data pages start with 0x40 (len 64) ...
    first = bytes(img).find(b"\x40abcdefgh")
    assert first > 0
    img[first] = 0xff
    img[first + 1] = 0xff
    img[first + 2] = 0xff
    img[first + 3] = 0xff
    img[first + 4] = 0x0f  # 4 GiB
    t = ctx.open_image(bytes(img))
    try:
        with pytest.raises(E.EvqlError) as ei:
            t.query(Plan(S, select=[col("s"), count(1)], group_by=[col("s")])).run()
        assert ei.value.code == K.EVQL_EIO and "end of column" in ei.value.msg
    finally:
        t.close()


def test_string_keys_run_on_the_tables_dictionary(ctx):
    """one STRING key that is only grouped by and selected: the kernels group by the
    column's dictionary codes (string_dict.cc) -- exact 32-bit identities, no hashing in
    the row function -- and everything downstream (emission, ORDER BY, PARTIAL rows, the
    first-row strings) still sees the plan's string key.  Plans that read the column
    elsewhere, or other first-row values, keep the hashed identity."""
    from eventql_amd.plan import Order
    rng = np.random.default_rng(5)
    n = 120_000
    u = rng.integers(0, 40_000, n)
    vals = [b"" if k % 97 == 0 else (b"key/%d" % k if k % 3 else b"k%dx" % k) * (1 + k % 4) for k in u]
    present = (rng.random(n) >= 0.05).astype(np.uint8)
    for img in (build(vals), build(vals, present=present)):
        t = ctx.open_image(img)
        s, x = col("s"), col("x")
        coded = [Plan(S, select=[s, count(1), sum_(x)], group_by=[s]),
                 Plan(S, select=[count(1), s, sum_(x % 7)], group_by=[s], where=Call("eq", x % 3, lit(1))),
                 Plan(S, select=[s, count(1)], group_by=[s], groups_hint=100_000)]
        hashed = [Plan(S, select=[s, x, count(1)], group_by=[s]),                 # first row of x
                  Plan(S, select=[s, count(1)], group_by=[s], where=s >= "key/2"),  # bytes in WHERE
                  Plan(S, select=[s, x % 2, count(1)], group_by=[s, x % 2])]       # two keys
        for plan, want_dict in [(p, True) for p in coded] + [(p, False) for p in hashed]:
            exp = O.oracle_run(img, plan)
            q = t.query(plan)
            src = q.kernel_source()
            assert ("evql_ident_add" not in src) == want_dict, plan.select
            got = q.run()
            assert got.nrows == exp.nrows
            # (first-row values of non-key columns are compared too: one table, one scan order)
            ki = [i for i, p in enumerate(plan.select) if p.return_type == K.T_STRING][0]
            if len(plan.group) == 2:
                key = lambda r: (r[0], r[1])
            else:
                key = lambda r: r[ki]
            assert sorted(got.rows(), key=lambda r: repr(key(r))) == \
                sorted(exp.rows(), key=lambda r: repr(key(r)))
            q.close()
        # PARTIAL rows of a coded plan: SHA1 keys + states, byte for byte
        pp = Plan(S, select=[s, count(1), sum_(x)], group_by=[s], mode=K.MODE_PARTIAL)
        q = t.query(pp)
        assert "evql_ident_add" not in q.kernel_source()
        got = sorted(q.run().rows())
        q.close()
        ep = O.oracle_run(img, pp)
        want = sorted((ep.keys[20 * i:20 * i + 20], ep.columns[0][i]) for i in range(ep.nrows))
        assert got == want
        # ORDER BY count desc, then the string key, LIMIT: over the translated records
        p2 = Plan(S, select=[s, count(1)], group_by=[s])
        order = Order(p2, [(1, True), (0, False)], limit=9, offset=1)
        q = t.query(p2)
        q.set_order(order)
        got = q.run().rows()
        q.close()
        assert got == O.oracle_run(img, p2, order=order).rows()
        t.close()
