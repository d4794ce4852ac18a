"""shared synthetic tables for the tests (built with the product's writer; the
writer itself is pinned against the reference reader in test_format_cpu.py)"""
import functools
import os

import numpy as np

import eventql_amd as E
from eventql_amd import capi as K, synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

SURVEY_SCHEMA = dict(k=K.T_UINT64, v=K.T_FLOAT64, a=K.T_UINT64, b=K.T_UINT64,
                     n=K.T_UINT64, s=K.T_STRING)


def survey_columns(n):
    c = synth.table_columns(n)
    i = np.arange(n, dtype=np.uint64)
    c["n"] = i
    c["n_present"] = (i % 3 != 0).astype(np.uint8)
    c["s"] = [b"g%d" % kk for kk in c["k"]]
    return c


@functools.lru_cache(maxsize=4)
def survey_table(n=1_000_000):
    """the table of SURVEY.md 8c(ii) with the encodings the survey used:
    k LEB128, v IEEE754, a UINT32_BITPACKED(32), b UINT64_PLAIN, n nullable
    LEB128, s STRING_PLAIN.  Returns (image bytes, columns dict)."""
    c = survey_columns(n)
    w = E.Writer([
        dict(name="k", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_LEB128),
        dict(name="v", logical_type=K.COL_FLOAT, storage_type=K.ENC_FLOAT_IEEE754),
        dict(name="a", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT32_BITPACKED),
        dict(name="b", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_PLAIN),
        dict(name="n", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_LEB128,
             dlevel_max=1),
        dict(name="s", logical_type=K.COL_STRING, storage_type=K.ENC_STRING_PLAIN)])
    w.put("k", c["k"])
    w.put("v", c["v"])
    w.put("a", c["a"])
    w.put("b", c["b"])
    w.put("n", c["n"], present=c["n_present"])
    w.put("s", c["s"])
    w.commit(n)
    img = w.image()
    w.close()
    return img, c


MIXED_SCHEMA = dict(k=K.T_UINT64, v=K.T_FLOAT64, a=K.T_UINT64, b=K.T_UINT64, n=K.T_UINT64,
                    s=K.T_STRING, p=K.T_UINT64, f=K.T_BOOL, k10=K.T_UINT64, t=K.T_TIMESTAMP64,
                    nv=K.T_FLOAT64, nb=K.T_UINT64, w=K.T_UINT64, ns=K.T_STRING)


@functools.lru_cache(maxsize=4)
def mixed_table(n=300_000):
    """every encoding, nullable variants, a wide (multi-byte LEB128) column"""
    c = survey_columns(n)
    i = np.arange(n, dtype=np.uint64)
    c["p"] = c["a"] * np.uint64(3)
    c["f"] = c["a"] & np.uint64(1)
    c["k10"] = c["k"]
    c["t"] = np.uint64(1438055327000000) + i * np.uint64(1000003)
    c["nv"] = c["v"]
    c["nv_present"] = (c["b"] % np.uint64(5) != 0).astype(np.uint8)
    c["nb"] = c["b"]
    c["nb_present"] = (c["a"] % np.uint64(7) != 0).astype(np.uint8)
    c["w"] = c["x"]  # full 64-bit values: 10-byte LEB128s straddling pages
    c["ns"] = [b"s%d" % (kk % 50) for kk in c["k"]]
    c["ns_present"] = (i % 4 != 1).astype(np.uint8)
    w = E.Writer([
        dict(name="k", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_LEB128),
        dict(name="v", logical_type=K.COL_FLOAT, storage_type=K.ENC_FLOAT_IEEE754),
        dict(name="a", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT32_BITPACKED),
        dict(name="b", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_PLAIN),
        dict(name="n", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_LEB128,
             dlevel_max=1),
        dict(name="s", logical_type=K.COL_STRING, storage_type=K.ENC_STRING_PLAIN),
        dict(name="p", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT32_PLAIN),
        dict(name="f", logical_type=K.COL_BOOLEAN, storage_type=K.ENC_BOOLEAN_BITPACKED),
        dict(name="k10", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT32_BITPACKED,
             bitpack_max_value=1023),
        dict(name="t", logical_type=K.COL_DATETIME, storage_type=K.ENC_UINT64_LEB128),
        dict(name="nv", logical_type=K.COL_FLOAT, storage_type=K.ENC_FLOAT_IEEE754, dlevel_max=1),
        dict(name="nb", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT32_BITPACKED,
             dlevel_max=1),
        dict(name="w", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_LEB128),
        dict(name="ns", logical_type=K.COL_STRING, storage_type=K.ENC_STRING_PLAIN, dlevel_max=1)])
    for name in ("k", "v", "a", "b", "s", "p", "f", "k10", "t", "w"):
        w.put(name, c[name])
    for name in ("n", "nv", "nb", "ns"):
        w.put(name, c[name], present=c[name + "_present"])
    w.commit(n)
    img = w.image()
    w.close()
    return img, c


def compare_results(got_rows, exp_rows, types, key_cols=1, rel=1e-6, abs_tol=0.0):
    """order-insensitive comparison: integers / strings / NULLs bit-exact, floats
    within `rel` relative (BASELINE.json north_star: 1e-6)"""
    g = {tuple(r[:key_cols]): r for r in got_rows}
    e = {tuple(r[:key_cols]): r for r in exp_rows}
    assert len(g) == len(got_rows), "duplicate keys in result"
    assert set(g) == set(e), "group sets differ: %d vs %d" % (len(g), len(e))
    for k, er in e.items():
        gr = g[k]
        for ci, (a, b) in enumerate(zip(gr, er)):
            if types[ci] == K.T_FLOAT64 and a is not None and b is not None:
                if b != b:  # NaN
                    assert a != a, (k, gr, er)
                else:
                    if b in (float("inf"), float("-inf")):
                        assert a == b, (k, ci, gr, er)
                    else:
                        assert abs(a - b) <= rel * max(abs(b), 1e-300) + abs_tol, (k, ci, gr, er)
            else:
                assert a == b, (k, ci, gr, er)


# ---- full-range columns: bit-packed / plain / LEB128 values next to 2^17 .. 2^64 ------------
RANGES_SCHEMA = dict(x17=K.T_UINT64, x24=K.T_UINT64, x31=K.T_UINT64, x32=K.T_UINT64,
                     p32=K.T_UINT64, q64=K.T_UINT64, l64=K.T_UINT64, g=K.T_UINT64,
                     fv=K.T_FLOAT64, nx24=K.T_UINT64)
RANGES = dict(uint_cols=["x17", "x24", "x31", "x32", "p32", "q64", "l64", "g", "nx24"],
              float_cols=["fv"], bool_cols=[], key_cols=["g", "x17", "nx24"],
              first_cols=["x24", "q64", "fv"],
              lits=[1, 3, 13, 255, 65536, (1 << 24) - 1, (1 << 31) + 5, (1 << 32) - 1, 1 << 63])


@functools.lru_cache(maxsize=2)
def ranges_table(n=200_000):
    rng = np.random.default_rng(77)

    def edgy(bits):
        # uniform, plus a third of the rows within 40 of the top of the range
        top = (1 << bits) - 1
        v = rng.integers(0, top, n, dtype=np.uint64, endpoint=True)
        near = np.uint64(top) - rng.integers(0, 40, n, dtype=np.uint64)
        return np.where(rng.random(n) < 0.33, near, v).astype(np.uint64)

    c = dict(x17=edgy(17), x24=edgy(24), x31=edgy(31), x32=edgy(32), p32=edgy(32),
             q64=edgy(64), l64=edgy(64), g=rng.integers(0, 37, n, dtype=np.uint64),
             fv=rng.normal(0, 1e6, n), nx24=edgy(24))
    pres = (rng.random(n) < 0.8).astype(np.uint8)
    bp = lambda name, bits, **kw: dict(name=name, logical_type=K.COL_UNSIGNED_INT,
                                       storage_type=K.ENC_UINT32_BITPACKED,
                                       bitpack_max_value=(1 << bits) - 1, **kw)
    w = E.Writer([bp("x17", 17), bp("x24", 24), bp("x31", 31), bp("x32", 32),
                  dict(name="p32", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT32_PLAIN),
                  dict(name="q64", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_PLAIN),
                  dict(name="l64", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_LEB128),
                  bp("g", 6),
                  dict(name="fv", logical_type=K.COL_FLOAT, storage_type=K.ENC_FLOAT_IEEE754),
                  bp("nx24", 24, dlevel_max=1)])
    for name in ("x17", "x24", "x31", "x32", "p32", "q64", "l64", "g", "fv"):
        w.put(name, c[name])
    w.put("nx24", c["nx24"], present=pres)
    w.commit(n)
    img = w.image()
    w.close()
    return img
