"""Device-side cstable writer (evql_table_from_device_columns) against the host
writer (evql_writer_*, itself validated against the reference's reader and writer) and
against the reference's own CSTableWriter (oracle/_ref): the file bytes -- in both page
placement orders a sequential writer produces --, the decoded (d, value) streams through
the oracle's and the reference's reader, and query results over the written table."""
import os

import numpy as np
import pytest
import torch

import eventql_amd as E
from eventql_amd import capi as K
from eventql_amd.plan import Plan, col, count, sum_, min_, max_
import oracle_lib as O
import tables as T

pytestmark = pytest.mark.gpu

U, F, B = K.COL_UNSIGNED_INT, K.COL_FLOAT, K.COL_BOOLEAN
REQUIRED = [
    dict(name="p64", logical_type=U, storage_type=K.ENC_UINT64_PLAIN),
    dict(name="f", logical_type=F, storage_type=K.ENC_FLOAT_IEEE754),
    dict(name="p32", logical_type=U, storage_type=K.ENC_UINT32_PLAIN),
    dict(name="bp10", logical_type=U, storage_type=K.ENC_UINT32_BITPACKED, bitpack_max_value=1023),
    dict(name="bp32", logical_type=U, storage_type=K.ENC_UINT32_BITPACKED),
    dict(name="bp17", logical_type=U, storage_type=K.ENC_UINT32_BITPACKED,
         bitpack_max_value=(1 << 17) - 1),
    dict(name="flag", logical_type=B, storage_type=K.ENC_BOOLEAN_BITPACKED),
    dict(name="leb", logical_type=U, storage_type=K.ENC_UINT64_LEB128),
]
OPTIONAL = [dict(c, name="o_" + c["name"], dlevel_max=1) for c in REQUIRED]


def make_columns(n, seed):
    rng = np.random.default_rng(seed)
    c = {
        "p64": rng.integers(0, 1 << 63, n, dtype=np.uint64) * np.uint64(2) + np.uint64(1),
        "f": rng.normal(0, 1e6, n).view(np.uint64),
        "p32": rng.integers(0, 1 << 32, n, dtype=np.uint64),
        "bp10": rng.integers(0, 1024, n, dtype=np.uint64),
        "bp32": rng.integers(0, 1 << 32, n, dtype=np.uint64),
        "bp17": rng.integers(0, 1 << 17, n, dtype=np.uint64),
        "flag": rng.integers(0, 2, n, dtype=np.uint64),
        # every LEB128 length, 1 .. 10 bytes
        "leb": (rng.integers(0, 1 << 63, n, dtype=np.uint64) * np.uint64(2) + np.uint64(1)) >>
               rng.integers(0, 64, n).astype(np.uint64),
    }
    for name in list(c):
        c["o_" + name] = c[name].copy()
        c["o_" + name + "_null"] = (rng.random(n) < 0.3).astype(np.uint8)
    return c


def host_image(specs, c, n):
    w = E.Writer(specs)
    for s in specs:
        name = s["name"]
        v = c[name].view(np.float64) if s["logical_type"] == F else c[name]
        pres = None
        if s.get("dlevel_max", 0):
            pres = (1 - c[name + "_null"]).astype(np.uint8)
        w.put(name, v, present=pres)
    w.commit(n)
    img = w.image()
    w.close()
    return img


def device_table(ctx, specs, c, n):
    keep, vals, nulls = [], {}, {}
    for s in specs:
        name = s["name"]
        tv = torch.from_numpy(c[name].view(np.int64).copy()).cuda()
        keep.append(tv)
        vals[name] = tv.data_ptr() if n else None
        if s.get("dlevel_max", 0):
            tn = torch.from_numpy(c[name + "_null"].copy()).cuda()
            keep.append(tn)
            # (an empty tensor has no storage: any non-null address will do for n == 0)
            nulls[name] = tn.data_ptr() if n else 1
    torch.cuda.synchronize()
    t = ctx.table_from_device_columns(specs, vals, nulls, n)
    del keep
    return t


@pytest.mark.parametrize("n", [0, 1, 127, 128, 129, 2049, 65536, 65537, 131072, 131073, 400_001])
def test_required_columns_are_byte_identical_to_the_host_writer(ctx, n):
    c = make_columns(n, 100 + n)
    t = device_table(ctx, REQUIRED, c, n)
    assert t.num_rows == n
    dev = t.download_image()
    t.close()
    host = host_image(REQUIRED, c, n)
    assert len(dev) == len(host)
    assert dev == host


@pytest.mark.parametrize("n", [0, 1, 129, 5000, 131073, 700_001])
def test_optional_columns_round_trip(ctx, n, tmp_path):
    c = make_columns(n, 200 + n)
    specs = OPTIONAL + REQUIRED[:1]
    t = device_table(ctx, specs, c, n)
    dev = t.download_image()
    host = host_image(specs, c, n)
    # page for page in the order a column-after-column writer allocates them
    assert dev == host
    path = str(tmp_path / "dev.cst")
    with open(path, "wb") as f:
        f.write(dev)
    for kind in (("orc", "ref") if O.have_ref() else ("orc",)):
        r = O.TableReader(path, kind)
        assert {x["name"] for x in r.columns()} == {s["name"] for s in specs}
        for s in OPTIONAL:
            name = s["name"]
            is_float = s["logical_type"] == F
            rl, dl, pr, v = r.read(name, n, "float" if is_float else "uint")
            null = c[name + "_null"].astype(bool)
            assert (np.asarray(dl) == (1 - c[name + "_null"])).all()
            exp = c[name].copy()
            got = (np.asarray(v, np.float64).view(np.uint64) if is_float
                   else np.asarray(v, np.uint64))
            assert (got[~null] == exp[~null]).all(), name
        r.close()
    # and the operator reads the device-written table like the host-written one
    if n:
        S = {"o_bp10": K.T_UINT64, "o_p64": K.T_UINT64, "o_leb": K.T_UINT64, "o_f": K.T_FLOAT64,
             "p64": K.T_UINT64}
        plan = Plan(S, select=[col("o_bp10"), count(1), sum_(col("o_leb")), min_(col("o_f")),
                               max_(col("o_p64")), sum_(col("p64"))], group_by=[col("o_bp10")])
        exp = O.oracle_run(host, plan)
        q = t.query(plan)
        T.compare_results(q.run().rows(), exp.rows(), exp.types, key_cols=1)
        q.close()
    t.close()


def test_unaligned_input_arrays(ctx):
    """value / NULL-flag arrays at odd addresses take the scalar load paths"""
    n = 70_001
    c = make_columns(n + 1, 7)
    specs = [OPTIONAL[0], OPTIONAL[7], REQUIRED[7]]
    tv = {s["name"]: torch.from_numpy(c[s["name"]].view(np.int64).copy()).cuda() for s in specs}
    tn = {s["name"]: torch.from_numpy(c[s["name"] + "_null"].copy()).cuda() for s in specs[:2]}
    torch.cuda.synchronize()
    t = ctx.table_from_device_columns(specs, {k: v[1:].data_ptr() for k, v in tv.items()},
                                      {k: v[1:].data_ptr() for k, v in tn.items()}, n)
    dev = t.download_image()
    t.close()
    cs = {k: v[1:] for k, v in c.items()}
    assert dev == host_image(specs, cs, n)


def test_bad_arguments(ctx):
    x = torch.zeros(16, dtype=torch.int64, device="cuda")
    # a repeated column without its level arrays
    spec = dict(name="r", logical_type=U, storage_type=K.ENC_UINT64_PLAIN, rlevel_max=1, dlevel_max=1)
    with pytest.raises(E.EvqlError) as ei:
        ctx.table_from_device_columns([spec], {"r": x.data_ptr()}, {}, 16)
    assert ei.value.code == K.EVQL_EARG
    # NULL flags without an optional column (and the other way round)
    spec = dict(name="a", logical_type=U, storage_type=K.ENC_UINT64_PLAIN)
    with pytest.raises(E.EvqlError) as ei:
        ctx.table_from_device_columns([spec], {"a": x.data_ptr()}, {"a": x.data_ptr()}, 16)
    assert ei.value.code == K.EVQL_EARG


def _string_inputs(vals):
    """(packed (len << 40 | offset) words, heap bytes) of a list of byte strings"""
    lens = np.array([len(v) for v in vals], dtype=np.uint64)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64) if len(vals) else lens
    heap = np.frombuffer(b"".join(vals) or b"\0", dtype=np.uint8).copy()
    return (lens << np.uint64(40)) | offs, heap


@pytest.mark.parametrize("n", [0, 1, 2047, 2049, 70_001, 300_000])
def test_string_columns_are_byte_identical_to_the_host_writer(ctx, n):
    """LenencStringPageWriter on the device: required and optional STRING_PLAIN columns,
    header widths 1..3 bytes, values straddling 512 KiB pages"""
    rng = np.random.default_rng(900 + n)
    lens = rng.choice([0, 1, 5, 9, 127, 128, 300, 20000], n, p=[.1, .1, .4, .3, .04, .03, .02, .01])
    vals = [bytes(rng.integers(0, 256, int(L), dtype=np.uint8)) for L in lens]
    null = (rng.random(n) < 0.25).astype(np.uint8)
    specs = [dict(name="s", logical_type=K.COL_STRING, storage_type=K.ENC_STRING_PLAIN),
             dict(name="os", logical_type=K.COL_STRING, storage_type=K.ENC_STRING_PLAIN, dlevel_max=1),
             dict(name="x", logical_type=U, storage_type=K.ENC_UINT64_PLAIN)]
    words, heap = _string_inputs(vals)
    tw = torch.from_numpy(words.view(np.int64).copy()).cuda()
    th = torch.from_numpy(heap).cuda()
    tn = torch.from_numpy(null.copy()).cuda()
    tx = torch.arange(n, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    one = 1
    t = ctx.table_from_device_columns(
        specs, {"s": tw.data_ptr() if n else None, "os": tw.data_ptr() if n else None,
                "x": tx.data_ptr() if n else None},
        {"os": tn.data_ptr() if n else one}, n, heaps={"s": th.data_ptr(), "os": th.data_ptr()})
    dev = t.download_image()
    w = E.Writer(specs)
    w.put("s", vals)
    w.put("os", vals, present=(1 - null).astype(np.uint8))
    w.put("x", np.arange(n, dtype=np.uint64))
    w.commit(n)
    host = w.image()
    w.close()
    assert dev == host
    # the table written on the device reads like the host-written one, strings included
    if n:
        S = dict(s=K.T_STRING, os=K.T_STRING, x=K.T_UINT64)
        for plan in (Plan(S, select=[col("x"), col("s"), col("os"), count(1)], group_by=[col("x")],
                          groups_hint=2 * n),
                     Plan(S, select=[col("os"), count(1), sum_(col("x"))], group_by=[col("os")])):
            exp = O.oracle_run(host, plan)
            q = t.query(plan)
            T.compare_results(q.run().rows(), exp.rows(), exp.types, key_cols=1)
            q.close()
            if n > 2049:
                exp2 = O.oracle_run(dev, plan)   # and the oracle reads the device-written file
                T.compare_results(exp2.rows(), exp.rows(), exp.types, key_cols=1)
    t.close()


@pytest.mark.parametrize("nrec", [0, 1, 300, 40_000, 200_000])
def test_repeated_columns(ctx, nrec):
    """REPEATED RECORD items{position, price, tag}: the (r, d, value) triples of every
    slot go in as level arrays + values (ColumnWriter::write*(r, d, v)); the file must
    read back slot for slot, equal the host writer's bytes while every stream fits one
    page, and answer nested scans like the host-written table"""
    rng = np.random.default_rng(40 + nrec)
    cnt = np.minimum(rng.geometric(0.35, nrec) - 1, 8)
    slots = np.maximum(cnt, 1)
    total = int(slots.sum())
    starts = np.concatenate([[0], np.cumsum(slots)[:-1]]).astype(np.int64) if nrec else np.zeros(0, np.int64)
    rl = np.ones(total, np.uint8)
    rl[starts] = 0
    rec_of_slot = np.repeat(np.arange(nrec), slots)
    dl = np.where(cnt[rec_of_slot] > 0, 2, rng.integers(0, 2, total)).astype(np.uint8)
    pos = (np.arange(total) - starts[rec_of_slot] + 1).astype(np.uint64)
    price = rng.integers(1, 1 << 40, total).astype(np.uint64)
    tags = [b"t%d" % (x % 17) for x in price]
    ids = np.arange(nrec, dtype=np.uint64) * np.uint64(7)
    specs = [dict(name="id", logical_type=U, storage_type=K.ENC_UINT64_PLAIN),
             dict(name="items.position", logical_type=U, storage_type=K.ENC_UINT32_BITPACKED,
                  rlevel_max=1, dlevel_max=2, bitpack_max_value=15),
             dict(name="items.price", logical_type=U, storage_type=K.ENC_UINT64_LEB128,
                  rlevel_max=1, dlevel_max=2),
             dict(name="items.tag", logical_type=K.COL_STRING, storage_type=K.ENC_STRING_PLAIN,
                  rlevel_max=1, dlevel_max=2)]
    w = E.Writer(specs)
    w.put("id", ids)
    w.put("items.position", pos, rlvl=rl.astype(np.uint64), dlvl=dl.astype(np.uint64))
    w.put("items.price", price, rlvl=rl.astype(np.uint64), dlvl=dl.astype(np.uint64))
    w.put("items.tag", tags, rlvl=rl.astype(np.uint64), dlvl=dl.astype(np.uint64))
    w.commit(nrec)
    host = w.image()
    w.close()
    words, heap = _string_inputs(tags)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64 if a.dtype.itemsize == 8 else a.dtype).copy()).cuda()
    t_id, t_pos, t_price, t_w = dev(ids), dev(pos), dev(price), dev(words)
    t_rl, t_dl = torch.from_numpy(rl.copy()).cuda(), torch.from_numpy(dl.copy()).cuda()
    t_h = torch.from_numpy(heap).cuda()
    torch.cuda.synchronize()
    p = lambda tt: tt.data_ptr() if tt.numel() else None
    lv = (p(t_rl), p(t_dl), total)
    t = ctx.table_from_device_columns(
        specs, {"id": p(t_id), "items.position": p(t_pos), "items.price": p(t_price),
                "items.tag": p(t_w)}, None, nrec, heaps={"items.tag": t_h.data_ptr()},
        levels={"items.position": lv, "items.price": lv, "items.tag": lv})
    devimg = t.download_image()
    assert devimg == host
    if nrec:
        S = {"id": K.T_UINT64, "items.position": K.T_UINT64, "items.price": K.T_UINT64,
             "items.tag": K.T_STRING}
        for kw in (dict(select=[col("items.position"), count(1), sum_(col("items.price")), sum_(col("id"))],
                        group_by=[col("items.position")]),
                   dict(select=[col("items.tag"), count(1), max_(col("items.price"))],
                        group_by=[col("items.tag")])):
            plan = Plan(S, scan_mode=K.SCAN_NESTED, **kw)
            exp = O.oracle_run(host, plan)
            q = t.query(plan)
            T.compare_results(q.run().rows(), exp.rows(), exp.types, key_cols=1)
            q.close()
            exp2 = O.oracle_run(devimg, plan)
            T.compare_results(exp2.rows(), exp.rows(), exp.types, key_cols=1)
    t.close()


def _ref_schema(specs):
    """flat / one-level nested specs as the reference TableSchema node list"""
    nodes, parents = [], {}
    for sp in specs:
        parts = sp["name"].split(".")
        parent = -1
        if len(parts) == 2:
            if parts[0] not in parents:
                parents[parts[0]] = len(nodes)
                nodes.append(dict(name=parts[0], type=K.COL_SUBRECORD, encoding=0, repeated=1,
                                  optional=0, parent=-1))
            parent = parents[parts[0]]
        nodes.append(dict(name=parts[-1], type=sp["logical_type"], encoding=sp["storage_type"],
                          repeated=0, optional=1 if sp.get("dlevel_max", 0) and parent < 0 else
                          (1 if sp.get("optional") else 0), parent=parent))
    return nodes


def _in_reference_header_order(specs, tmp_path):
    """the reference lists the columns of the file header in the order of its schema
    map, not of declaration: learn that order (and the column ids) from an empty file
    and use it as the declaration AND write order of both writers"""
    path = str(tmp_path / "order.cst")
    if os.path.exists(path):
        os.unlink(path)
    O.ref_write_table_by_records(path, _ref_schema(specs), [], 0)
    r = O.TableReader(path, "ref")
    hdr = r.columns()
    r.close()
    by_name = {s["name"]: s for s in specs}
    return [dict(by_name[h["name"]], column_id=h["column_id"]) for h in hdr]


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref (the reference's cstable library) not built")
@pytest.mark.parametrize("n", [1, 129, 131_073, 700_001])
def test_row_order_is_byte_identical_to_the_reference_writer(ctx, n, tmp_path):
    """EVQL_PAGE_ORDER_ROWS: the file the REFERENCE's CSTableWriter produces when a table
    is written row by row, every column per row (11 PLAIN64 pages, 6 bit-packed level
    pages and 5-7 LEB128 pages per stream at 700,001 rows, allocated interleaved)"""
    c = make_columns(n, 300 + n)
    bitpacked = {K.ENC_UINT32_BITPACKED, K.ENC_BOOLEAN_BITPACKED}
    # (TableSchema carries no bit-pack maximum: the reference writer uses the defaults)
    specs = [s for s in OPTIONAL + REQUIRED
             if s["storage_type"] not in bitpacked or "bitpack_max_value" not in s]
    declared = specs
    specs = _in_reference_header_order(declared, tmp_path)
    t = _device_table(ctx, specs, c, n, K.PAGE_ORDER_ROWS)
    dev = t.download_image()
    t.close()
    cols = []
    for s in specs:
        name = s["name"]
        pres = (1 - c[name + "_null"]).astype(np.uint8) if s.get("dlevel_max", 0) else None
        cols.append((name, "float" if s["logical_type"] == F else "uint", c[name], None, None, pres))
    path = str(tmp_path / "ref.cst")
    O.ref_write_table_by_records(path, _ref_schema(declared), cols, n)
    ref = open(path, "rb").read()
    assert len(dev) == len(ref)
    assert dev == ref
    # the column-after-column order is a different file
    if n > 131_072:
        t2 = _device_table(ctx, specs, c, n, K.PAGE_ORDER_COLUMNS)
        other = t2.download_image()
        t2.close()
        assert other != ref  # (index offsets are varuints: even the length may differ)


def _device_table(ctx, specs, c, n, order):
    keep, vals, nulls = [], {}, {}
    for s in specs:
        name = s["name"]
        tv = torch.from_numpy(c[name].view(np.int64).copy()).cuda()
        keep.append(tv)
        vals[name] = tv.data_ptr()
        if s.get("dlevel_max", 0):
            tn = torch.from_numpy(c[name + "_null"].copy()).cuda()
            keep.append(tn)
            nulls[name] = tn.data_ptr()
    torch.cuda.synchronize()
    t = ctx.table_from_device_columns(specs, vals, nulls, n, page_order=order)
    del keep
    return t


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref (the reference's cstable library) not built")
@pytest.mark.parametrize("nrec", [300, 200_000])
def test_row_order_of_repeated_columns(ctx, nrec, tmp_path):
    """record by record with a REPEATED RECORD: a level page of `items.*` is allocated
    by the record that holds slot 131,072 k, a data page by the record of its first value"""
    rng = np.random.default_rng(70 + nrec)
    cnt = np.minimum(rng.geometric(0.3, nrec) - 1, 9)
    slots = np.maximum(cnt, 1)
    total = int(slots.sum())
    starts = np.concatenate([[0], np.cumsum(slots)[:-1]]).astype(np.int64)
    rl = np.ones(total, np.uint8)
    rl[starts] = 0
    rec_of_slot = np.repeat(np.arange(nrec), slots)
    dl = np.where(cnt[rec_of_slot] > 0, 2, 1).astype(np.uint8)  # empty list: d = 1
    pos = (np.arange(total) - starts[rec_of_slot] + 1).astype(np.uint64)
    price = rng.integers(1, 1 << 50, total).astype(np.uint64)
    ids = np.arange(nrec, dtype=np.uint64) * np.uint64(7)
    score = rng.normal(0, 10, nrec)
    specs = [dict(name="id", logical_type=U, storage_type=K.ENC_UINT64_PLAIN),
             dict(name="items.position", logical_type=U, storage_type=K.ENC_UINT32_PLAIN,
                  rlevel_max=1, dlevel_max=2, optional=1),
             dict(name="items.price", logical_type=U, storage_type=K.ENC_UINT64_LEB128,
                  rlevel_max=1, dlevel_max=2, optional=1),
             dict(name="score", logical_type=F, storage_type=K.ENC_FLOAT_IEEE754)]
    declared = specs
    specs = _in_reference_header_order(declared, tmp_path)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64 if a.dtype.itemsize == 8 else a.dtype).copy()).cuda()
    t_id, t_pos, t_price, t_sc = dev(ids), dev(pos), dev(price), dev(score)
    t_rl, t_dl = torch.from_numpy(rl.copy()).cuda(), torch.from_numpy(dl.copy()).cuda()
    torch.cuda.synchronize()
    lv = (t_rl.data_ptr(), t_dl.data_ptr(), total)
    t = ctx.table_from_device_columns(
        specs, {"id": t_id.data_ptr(), "items.position": t_pos.data_ptr(),
                "items.price": t_price.data_ptr(), "score": t_sc.data_ptr()}, None, nrec,
        levels={"items.position": lv, "items.price": lv}, page_order=K.PAGE_ORDER_ROWS)
    devimg = t.download_image()
    t.close()
    r64, d64 = rl.astype(np.uint64), dl.astype(np.uint64)
    by_name = {"id": ("id", "uint", ids, None, None, None),
               "items.position": ("items.position", "uint", pos, r64, d64, None),
               "items.price": ("items.price", "uint", price, r64, d64, None),
               "score": ("score", "float", score, None, None, None)}
    cols = [by_name[sp["name"]] for sp in specs]
    path = str(tmp_path / "ref.cst")
    O.ref_write_table_by_records(path, _ref_schema(declared), cols, nrec)
    ref = open(path, "rb").read()
    assert len(devimg) == len(ref)
    assert devimg == ref
