#!/usr/bin/env python3
"""Soak on an MI355X box: random GROUP BYs over the LSM partitions of tests/lsm_tables.py --
the chain operator (evql_query_create_chain: device row filters, one launch per file,
chain merge) against the oracle's chain run (orc_query_run_chain, pinned on the reference's
PartitionCursor by tests/golden/ref_csql_lsm.json), FINAL and PARTIAL.  Other seeds than the
fixtures.  usage: tests/soak_chain_fuzz.py <first seed> <count>"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import eventql_amd as E  # noqa: E402
from eventql_amd import capi as K  # noqa: E402
from eventql_amd.plan import Plan, CompileError  # noqa: E402
import lsm_tables  # noqa: E402
import oracle_lib as O  # noqa: E402
import refcases  # noqa: E402
import sqlgen  # noqa: E402
import tables as T  # noqa: E402
from test_lsm_partition import oracle_filters, scan_order_images  # noqa: E402


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    ctx = E.Context(0)
    total = lowered = errors = 0
    bad = []
    for pname in sorted(lsm_tables.PARTITIONS):
        files = lsm_tables.partition(pname)
        tabs = [ctx.open_image(f[1]) for f in reversed(files)]
        ch = E.LsmChain(ctx)
        for t, f in zip(tabs, reversed(files)):
            ch.add(t, has_skiplist=f[2], has_updates=f[3])
        ch.build()
        imgs, filters = scan_order_images(pname), oracle_filters(pname)
        for seed in range(first, first + count):
            g = refcases.RefGen(seed, **refcases.LSM)
            g.count_cols = ["k", "a", "n", "v"]
            kw = g.plan_kwargs([1])
            for mode in (K.MODE_FINAL, K.MODE_PARTIAL):
                try:
                    plan = Plan(lsm_tables.LSM_SCHEMA, mode=mode, **kw)
                except CompileError:
                    continue
                total += 1
                try:
                    exp = O.oracle_run_chain(imgs, filters, plan)
                    exp_err = None
                except RuntimeError as e:
                    exp, exp_err = None, str(e)
                try:
                    q = ch.query(plan)
                except E.EvqlError as e:
                    if e.code != K.EVQL_ENOTSUP:
                        bad.append((pname, seed, mode, "create: " + e.msg))
                    continue
                lowered += 1
                try:
                    try:
                        got = q.run()
                    except E.EvqlError as e:
                        if exp_err is None or ("zero" in exp_err) != ("zero" in e.msg):
                            bad.append((pname, seed, mode, "error: " + e.msg, exp_err))
                        else:
                            errors += 1
                        continue
                    if exp_err is not None:
                        bad.append((pname, seed, mode, "no error, oracle: " + exp_err))
                        continue
                    if mode == K.MODE_PARTIAL:
                        e = {exp.keys[20 * i:20 * i + 20]: exp.columns[0][i] for i in range(exp.nrows)}
                        if dict(got.rows()) != e:
                            bad.append((pname, seed, mode, "partial rows differ"))
                    else:
                        try:
                            assert got.nrows == exp.nrows, (got.nrows, exp.nrows)
                            T.compare_results(got.rows(), exp.rows(), exp.types,
                                              key_cols=len(kw["group_by"]), rel=1e-6, abs_tol=1e-3)
                        except AssertionError as e:
                            bad.append((pname, seed, mode, "rows differ: " + str(e)[:200]))
                finally:
                    q.close()
        ch.close()
        for t in tabs:
            t.close()
        print("[chain soak] %s done (%d plans so far)" % (pname, total), flush=True)
    print(json.dumps(dict(plans=total, lowered=lowered, both_failed_alike=errors, mismatches=len(bad))))
    for b in bad[:10]:
        print("MISMATCH", b)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
