"""The benchmark query shapes of BASELINE.json (configs 2 and 3) as plans, plus
precompilation of their fused kernels into the in-tree kernel cache."""
from . import capi as K
from .plan import Plan, col, count, sum_

SCHEMA = dict(k=K.T_UINT64, a=K.T_UINT64, b=K.T_UINT64, v=K.T_FLOAT64, u=K.T_UINT64)

PLAIN_COLUMNS = [
    dict(name="k", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_PLAIN),
    dict(name="a", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_PLAIN),
    dict(name="b", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_PLAIN),
    dict(name="v", logical_type=K.COL_FLOAT, storage_type=K.ENC_FLOAT_IEEE754),
    dict(name="u", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_PLAIN),
]


def config2(groups_hint=1000, **kw):
    """SELECT k, sum(v), count(1) GROUP BY k   (16 B/row)"""
    return Plan(SCHEMA, select=[col("k"), sum_(col("v")), count(1)], group_by=[col("k")],
                groups_hint=groups_hint, **kw)


def config3(groups_hint=1000, **kw):
    """WHERE a > 30000 AND b < 30000 ; k, sum(v), count(1), sum(b) GROUP BY k  (32 B/row)"""
    return Plan(SCHEMA, select=[col("k"), sum_(col("v")), count(1), sum_(col("b"))],
                group_by=[col("k")], where=(col("a") > 30000) & (col("b") < 30000),
                groups_hint=groups_hint, **kw)


def config4(groups_hint=10_000_000, **kw):
    """high-cardinality GROUP BY on a 64-bit key column u, 3 aggregates"""
    return Plan(SCHEMA, select=[col("u"), sum_(col("a")), count(1), sum_(col("v"))],
                group_by=[col("u")], groups_hint=groups_hint, **kw)


STRING_KEY_SCHEMA = dict(s=K.T_STRING, a=K.T_UINT64, v=K.T_FLOAT64)
STRING_KEY_COLUMNS = [
    dict(name="s", logical_type=K.COL_STRING, storage_type=K.ENC_STRING_PLAIN),
    dict(name="a", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_PLAIN),
    dict(name="v", logical_type=K.COL_FLOAT, storage_type=K.ENC_FLOAT_IEEE754),
]


def config4s(groups_hint=10_000_000, **kw):
    """BASELINE.json configs[3] as written: high-cardinality GROUP BY on a STRING key
    ("g" + u, STRING_PLAIN; identity = two 64-bit hashes of the bytes), 3 aggregates"""
    return Plan(STRING_KEY_SCHEMA, select=[col("s"), sum_(col("a")), count(1), sum_(col("v"))],
                group_by=[col("s")], groups_hint=groups_hint, **kw)


def string_key_table(ctx, rows, n_keys, seed):
    """config-4 partition with a real string key, generated in HBM (torch RNG) and
    encoded by the device writer: s = "g" + u (u uniform in [0, n_keys)), a in
    [0, 65536), v in [0, 16384).  Returns the resident table."""
    import torch
    from . import synth
    g = torch.Generator(device="cuda")
    g.manual_seed(seed & 0x7fffffffffffffff)
    u = torch.randint(0, n_keys, (rows,), generator=g, device="cuda", dtype=torch.int64)
    a = torch.randint(0, 65536, (rows,), generator=g, device="cuda", dtype=torch.int64)
    v = torch.rand(rows, generator=g, device="cuda", dtype=torch.float64) * 16384.0
    words, heap = synth.device_string_keys(u)
    del u
    torch.cuda.synchronize()
    t = ctx.table_from_device_columns(
        STRING_KEY_COLUMNS, {"s": words.data_ptr(), "a": a.data_ptr(), "v": v.data_ptr()}, None,
        rows, heaps={"s": heap.data_ptr()})
    torch.cuda.synchronize()
    return t


def precompile_all():
    from . import compile_only
    for p in (config2(), config3(), config4()):
        compile_only(p, PLAIN_COLUMNS)
    compile_only(config4s(), STRING_KEY_COLUMNS)
