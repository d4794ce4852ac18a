#!/usr/bin/env python3
"""Scratch: global group (no keys) whose select list reads a string column of the first row."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import eventql_amd as E
from eventql_amd import capi as K
from eventql_amd.plan import Plan, Call, Col, Lit, Agg
import oracle_lib as O
import tables as T

img, _ = T.mixed_table(300_000)
ctx = E.Context(0)
t = ctx.open_image(img)
S = T.MIXED_SCHEMA
for sel in ([Col("ns"), Agg("count", Lit(1))],
            [Call("ucase", Col("ns")), Agg("count", Lit(1))],
            [Col("s"), Agg("count", Lit(1))],
            [Call("concat", Col("ns"), Lit("x")), Agg("count", Lit(1))]):
    for mode in (K.MODE_FINAL, K.MODE_PARTIAL):
        plan = Plan(S, select=sel, group_by=[], where=Call("lt", Col("k"), Lit(60)), mode=mode)
        exp = O.oracle_run(img, plan)
        print("oracle", exp.rows(), flush=True)
        q = t.query(plan)
        got = q.run()
        print("gpu   ", got.rows(), flush=True)
        q.close()
print("ok")
