"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports
every symbol include/evql_gpu.h declares, refuses to compute without a device,
lowers/compiles plans for gfx950, and reports non-lowerable plans the way the
adapter expects (EVQL_ENOTSUP => fall back to the CPU operators)."""
import ctypes as C
import os
import re

import pytest

import eventql_amd as E
from eventql_amd import capi as K, bench_plans as B
from eventql_amd.plan import Plan, col, count, sum_, min_, mean, If, CompileError, Agg, Call

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(built):
    hdr = open(os.path.join(ROOT, "include", "evql_gpu.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(evql_[a-z0-9_]+)\s*\(", hdr))
    names = {n for n in names if not n.endswith("_fn")}
    assert len(names) > 35
    L = C.CDLL(E.LIB_PATH)
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, missing


def test_no_cpu_fallback_without_device(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(E.EvqlError) as ei:
        E.Context(0)
    assert ei.value.code == K.EVQL_EDEVICE


def test_product_does_not_reference_the_oracle():
    """the oracle is test infrastructure: nothing under eventql_amd/ or include/
    may import, link or name it"""
    bad = []
    for base in ("eventql_amd", "include"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            if "_obj" in dp or "_kcache" in dp:
                continue
            for fn in fns:
                if not fn.endswith((".py", ".cc", ".h", ".hip", ".c", "Makefile")):
                    continue
                txt = open(os.path.join(dp, fn), errors="replace").read()
                if re.search(r"oracle_lib|liboracle|orc_query_run|libcstable_ref|oracle/", txt):
                    bad.append(os.path.join(dp, fn))
    assert not bad, bad


def _cols(*names, **over):
    d = {c["name"]: dict(c) for c in B.PLAIN_COLUMNS}
    out = []
    for n in names:
        c = d[n]
        c.update(over.get(n, {}))
        out.append(c)
    return out


def test_compile_benchmark_kernels_for_gfx950(built, tmp_path):
    for fn in (B.config2, B.config3, B.config4):
        size = E.compile_only(fn(), B.PLAIN_COLUMNS, cache_dir=str(tmp_path))
        assert size > 4000
    assert len(os.listdir(tmp_path)) == 3
    # second call is served from the on-disk cache
    assert E.compile_only(B.config3(), B.PLAIN_COLUMNS, cache_dir=str(tmp_path)) > 4000


def test_compile_expression_coverage(built, tmp_path):
    """every lowerable op family (SURVEY 8a op table) generates compilable HIP"""
    S = dict(B.SCHEMA)
    a, b, v, k = col("a"), col("b"), col("v"), col("k")
    plan = Plan(S, select=[k, sum_(If(a > b, a - b, b / (a + 1))), count(v), min_(v),
                           mean(a), sum_(v) + 1.0],
                group_by=[k], where=(((a + b) * 2 - 1) % 7 > 3) | ((~(v * 1.5 / 2.0 >= 10.0)) &
                                                                   a.neq(b)),
                groups_hint=1000)
    assert E.compile_only(plan, B.PLAIN_COLUMNS, cache_dir=str(tmp_path)) > 4000
    # multi-column key (hashed identity), global aggregate, bit-packed + u32 columns
    plan = Plan(S, select=[k, b, count(1)], group_by=[k, b], groups_hint=100000)
    assert E.compile_only(plan, B.PLAIN_COLUMNS, cache_dir=str(tmp_path)) > 4000
    # count_distinct (HBM pair set) under every key mode
    cd = lambda x: Agg("count_distinct", x)
    for kw in (dict(select=[k, cd(a), count(1), cd(b % 7)], group_by=[k]),
               dict(select=[k, b, cd(a)], group_by=[k, b]),
               dict(select=[cd(a + b)], where=a > 5)):
        assert E.compile_only(Plan(S, **kw), B.PLAIN_COLUMNS, cache_dir=str(tmp_path)) > 4000
    plan = Plan(S, select=[count(1), sum_(a)], where=a < 100)
    cols = _cols("k", "a", "b", "v", a=dict(storage_type=K.ENC_UINT32_BITPACKED, bits=17),
                 b=dict(storage_type=K.ENC_UINT32_PLAIN))
    assert E.compile_only(plan, cols, cache_dir=str(tmp_path)) > 4000


def test_compile_string_predicates(built, tmp_path):
    """eq / neq / lt / lte / gt / gte / cmp on string columns and literals lower to
    bytewise compares in the kernel; other string expressions stay ENOTSUP"""
    S = dict(k=K.T_UINT64, s=K.T_STRING, s2=K.T_STRING)
    cols = [dict(name="k", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_PLAIN),
            dict(name="s", logical_type=K.COL_STRING, storage_type=K.ENC_STRING_PLAIN),
            dict(name="s2", logical_type=K.COL_STRING, storage_type=K.ENC_STRING_PLAIN,
                 dlevel_max=1)]
    s, s2, k = col("s"), col("s2"), col("k")
    plan = Plan(S, select=[k, count(1), sum_(If(s2 >= "m", 1, 0))], group_by=[k],
                where=(s.eq("g5") | (s < s2)) & s2.neq("") & (Call("cmp", s, s2) < 1))
    assert E.compile_only(plan, cols, cache_dir=str(tmp_path)) > 4000
    plan = Plan(S, select=[s < "g5", count(1)], group_by=[s < "g5"])
    assert E.compile_only(plan, cols, cache_dir=str(tmp_path)) > 4000
    bad = Plan(S, select=[count(1)], where=If(k > 1, s, s2).eq("x"))
    with pytest.raises(E.EvqlError) as ei:
        E.compile_only(bad, cols)
    assert ei.value.code == K.EVQL_ENOTSUP


def test_compile_within_record_scan(built, tmp_path):
    """EVQL_SCAN_NESTED_WITHIN_RECORD (CSTableScan.cc:440-487): the operators above
    the scan read the per-record aggregates as `$i`; only bare count / integer sum
    over a column or literal are lowered"""
    from eventql_amd.plan import out
    S = {"id": K.T_UINT64, "items.position": K.T_UINT64, "items.price": K.T_UINT64}
    cols = [dict(name="id", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_PLAIN),
            dict(name="items.position", logical_type=K.COL_UNSIGNED_INT,
                 storage_type=K.ENC_UINT32_PLAIN, rlevel_max=1, dlevel_max=1),
            dict(name="items.price", logical_type=K.COL_UNSIGNED_INT,
                 storage_type=K.ENC_UINT64_PLAIN, rlevel_max=1, dlevel_max=1)]
    inner = [count(col("items.position")), sum_(col("items.price")), count(1), sum_(col("id"))]
    plan = Plan(S, scan_select=inner, select=[out(0), count(1), sum_(out(1)), sum_(out(3))],
                group_by=[out(0)], scan_mode=K.SCAN_NESTED_WITHIN_RECORD)
    assert E.compile_only(plan, cols, cache_dir=str(tmp_path)) > 4000
    for bad in (dict(scan_select=[sum_(col("items.price") + 1)], select=[sum_(out(0))]),
                dict(scan_select=[min_(col("items.price"))], select=[sum_(out(0))]),
                dict(scan_select=[count(col("id")) + 1], select=[sum_(out(0))]),
                dict(scan_select=[count(1)], select=[sum_(out(0))], where=col("id") > 3)):
        with pytest.raises(E.EvqlError) as ei:
            E.compile_only(Plan(S, scan_mode=K.SCAN_NESTED_WITHIN_RECORD, **bad), cols)
        assert ei.value.code == K.EVQL_ENOTSUP


def test_not_lowerable_plans_are_reported(built):
    S = dict(B.SCHEMA)
    # (count_distinct in a partial aggregate is lowered: its saved state is the value set)
    p = Plan(S, select=[col("k"), Agg("count_distinct", col("a"))], group_by=[col("k")],
             mode=K.MODE_PARTIAL)
    E.compile_only(p, B.PLAIN_COLUMNS)
    # an aggregate id outside the table
    p = Plan(S, select=[col("k"), count(1)], group_by=[col("k")])
    p.select[1].struct.aggregate_fn = 99
    p._select[1].aggregate_fn = 99
    with pytest.raises(E.EvqlError) as ei:
        E.compile_only(p, B.PLAIN_COLUMNS)
    assert ei.value.code == K.EVQL_ENOTSUP
    # INT64 scan column: the reference answers EARG "illegal column type: INT64"
    S2 = dict(S)
    S2["a"] = K.T_INT64
    p = Plan(S2, select=[count(1)], where=col("a") > -1)
    with pytest.raises(E.EvqlError) as ei:
        E.compile_only(p, B.PLAIN_COLUMNS)
    assert ei.value.code == K.EVQL_EARG and "INT64" in ei.value.msg
    # unknown column
    p = Plan(dict(zz=K.T_UINT64), select=[count(1)], where=col("zz") > 1)
    with pytest.raises(E.EvqlError) as ei:
        E.compile_only(p, B.PLAIN_COLUMNS)
    assert ei.value.code == K.EVQL_EARG


def test_plan_builder_matches_reference_compiler_layout():
    """bytecode layout of compiler.cc:50-248"""
    S = dict(a=K.T_UINT64, k=K.T_UINT64)
    p = Plan(S, select=[sum_(If(col("a") > 5, 1, 0)) + 1], group_by=[col("k")])
    prog = p.select[0]
    ops = [(i.op, i.arg0) for i in prog.code]
    # method_call: GET; LITERAL 1; CALL add; RETURN
    assert ops[0] == (K.X_CALL_INSTANCE, K.INSTANCE_GET)
    assert ops[1][0] == K.X_LITERAL and ops[2] == (K.X_CALL_PURE, K.FN(K.FAM_ADD, K.TS_UINT64))
    assert ops[3][0] == K.X_RETURN
    acc = prog.struct.method_accumulate
    assert acc == 4
    # accumulate: a; 5; gt; CJUMP ->T; 0; JUMP ->E; T: 1; ACCUMULATE; RETURN
    kinds = [o[0] for o in ops[acc:]]
    assert kinds == [K.X_INPUT, K.X_LITERAL, K.X_CALL_PURE, K.X_CJUMP, K.X_LITERAL, K.X_JUMP,
                     K.X_LITERAL, K.X_CALL_INSTANCE, K.X_RETURN]
    cj = ops[acc + 3]
    assert cj[1] == acc + 6           # true branch starts after the JUMP
    assert ops[acc + 5][1] == acc + 7  # JUMP -> end of IF
    # count(x) inserts to_nil; literals type by sign / dot
    p = Plan(S, select=[count(col("a"))])
    assert [i.op for i in p.select[0].code][2:5] == [K.X_INPUT, K.X_CALL_PURE, K.X_CALL_INSTANCE]
    with pytest.raises(CompileError):
        Plan(S, select=[count(1)], where=col("a") > 1.5)  # gt<uint64,float64> is a type error
    with pytest.raises(CompileError):
        Plan(S, select=[Agg("max_by", col("a"))])


def _compile_into(d):
    import eventql_amd as E2
    from eventql_amd import bench_plans as B2
    return E2.compile_only(B2.config2(), B2.PLAIN_COLUMNS, d)


def test_kernel_cache_survives_damage_and_concurrent_writers(built, tmp_path):
    """one process per GPU compiles the same plan at the same time: every writer renames
    a file of its own into place; a truncated / foreign cache file is compiled again"""
    import glob
    import multiprocessing as mp
    d = str(tmp_path / "kc")
    os.makedirs(d)
    with mp.get_context("spawn").Pool(3) as pool:
        sizes = pool.map(_compile_into, [d] * 3)
    assert len(set(sizes)) == 1 and sizes[0] > 1000
    files = glob.glob(d + "/*")
    assert len(files) == 1 and files[0].endswith(".hsaco"), files   # no stray .tmp files
    assert os.path.getsize(files[0]) == sizes[0]
    with open(files[0], "r+b") as f:
        f.truncate(100)                     # an ELF header without its sections
    assert E.compile_only(B.config2(), B.PLAIN_COLUMNS, d) == sizes[0]
    assert os.path.getsize(files[0]) == sizes[0]
    with open(files[0], "wb") as f:
        f.write(b"not a code object")
    assert E.compile_only(B.config2(), B.PLAIN_COLUMNS, d) == sizes[0]
    assert os.path.getsize(files[0]) == sizes[0]
