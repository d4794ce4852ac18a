#!/usr/bin/env python3
"""Soak on an MI355X box: random GROUP BY plans over 3 hub ranks (threads, one GPU), both
exchange modes, against the oracle on the concatenated partitions.  Non-aggregate select
expressions other than the keys are dropped (any partition's first row is "the" first row).
usage: tests/soak_exchange_fuzz.py <first seed> <count>"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import eventql_amd as E  # noqa: E402
from eventql_amd import capi as K  # noqa: E402
from eventql_amd.plan import Plan, Agg, CompileError  # noqa: E402
import oracle_lib as O  # noqa: E402
import tables as T  # noqa: E402
import test_gpu_exchange as X  # noqa: E402
from test_gpu_fuzz import Gen  # noqa: E402

COLS = dict(uint_cols=["k", "a", "u"], float_cols=["v"], bool_cols=[], key_cols=["k", "s", "ns", "u"],
            first_cols=["a"], lits=[0, 1, 2, 7, 1000, 30000, 65535, 1 << 40])


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    parts = [X.partition(4000 + r, 30_000 + 3000 * r) for r in range(3)]
    img = X.image_of(parts)
    total = lowered = 0
    bad = []
    for seed in range(first, first + count):
        g = Gen(seed, **COLS)
        kw = g.plan_kwargs([1])
        for drop in ("row_end", "row_filter"):
            kw.pop(drop, None)
        nk = len(kw["group_by"])
        kw["select"] = kw["select"][:nk] + [e for e in kw["select"][nk:] if isinstance(e, Agg)]
        try:
            plan = Plan(X.S, **kw)
        except CompileError:
            continue
        total += 1
        try:
            exp = O.oracle_run(img, plan)
        except RuntimeError:
            continue  # (errors: every rank raises on its own rows or not -- not compared)
        for mode in (K.EXCHANGE_GATHER_ALL, K.EXCHANGE_BY_OWNER):
            try:
                res = X.run_ranks(3, parts, kw, mode)
            except AssertionError as e:
                msg = str(e)
                if "(status %d)" % K.EVQL_ENOTSUP in msg:
                    break  # (not lowerable: the CPU operators keep the plan)
                if "zero" in msg:
                    break  # (division by zero on some rank's rows)
                bad.append((seed, mode, "ranks failed: " + msg[:300]))
                break
            lowered += 1
            try:
                if mode == K.EXCHANGE_GATHER_ALL:
                    for rows, _, _ in res:
                        assert len(rows) == exp.nrows, (len(rows), exp.nrows)
                        T.compare_results(rows, exp.rows(), exp.types, key_cols=nk, rel=1e-6, abs_tol=1e-3)
                else:
                    union = [row for rows, _, _ in res for row in rows]
                    assert len(union) == exp.nrows, (len(union), exp.nrows)
                    T.compare_results(union, exp.rows(), exp.types, key_cols=nk, rel=1e-6, abs_tol=1e-3)
            except AssertionError as e:
                bad.append((seed, mode, "rows differ: " + str(e)[:300]))
        if (seed - first) % 20 == 19:
            print("[exchange soak] %d seeds done" % (seed - first + 1), flush=True)
    print(json.dumps(dict(plans=total, exchanges=lowered, mismatches=len(bad))))
    for b in bad[:10]:
        print("MISMATCH", b)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
