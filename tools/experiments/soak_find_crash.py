#!/usr/bin/env python3
"""Scratch: runs the soak's `strings` family one query per probe process (MODE gpu) and
prints the ones whose process dies.  usage: soak_find_crash.py <first> <count>"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.argv = [sys.argv[0]] + sys.argv[1:]
import refcases  # noqa: E402
import soak_reference_differential as S  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
cs = S.cases(first, count)["strings"]
img, _, kind = refcases.table_image("mixed")
with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "mixed.cst")
    open(path, "wb").write(img)
    for c in cs:
        cmds = ["TABLE t %s %s" % (path, kind), "ROWS on", "MODE gpu", "SQL " + c["sql"]]
        p = subprocess.run([S.PROBE], input="\n".join(cmds) + "\n", capture_output=True, text=True)
        if p.returncode != 0:
            print("CRASH rc=%d %s\n  %s\n  stderr: %s" % (p.returncode, c["id"], c["sql"], p.stderr[-500:]), flush=True)
print("done", len(cs))
