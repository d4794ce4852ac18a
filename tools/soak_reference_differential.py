#!/usr/bin/env python3
"""Differential soak on an MI355X box: random GROUP BY queries through the REFERENCE's own
engine (oracle/_ref/csql_probe, built from /root/reference by oracle/ref_csql/build.sh),
once with its CPU operators (MODE cpu) and once with the GPU operator plugged in (MODE gpu);
rows must be identical.  Same generators as the committed fixtures (tests/refcases.py),
other seeds.   usage: soak_reference_differential.py <first seed offset> <count>"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import refcases  # noqa: E402
import sqlgen  # noqa: E402
import tables as T  # noqa: E402
from refcases import RefGen, MIXED, MIXED_COUNT_COLS, _case  # noqa: E402

PROBE = os.path.join(ROOT, "oracle", "_ref", "csql_probe")


def cases(first, count):
    out = {"mixed": [], "ranges": []}
    for seed in range(first, first + count):
        g = RefGen(150_000 + seed, **MIXED)
        g.count_cols = MIXED_COUNT_COLS
        c = _case("mixed-s%d" % seed, "mixed", g.plan_kwargs([1]), T.MIXED_SCHEMA)
        if c:
            out["mixed"].append(c)
        g = RefGen(160_000 + seed, **T.RANGES)
        c = _case("ranges-s%d" % seed, "ranges", g.plan_kwargs([1]), T.RANGES_SCHEMA)
        if c:
            out["ranges"].append(c)
    return out


def run(mode, path, kind, sqls):
    cmds = ["TABLE t %s %s" % (path, kind), "ROWS on", "MODE " + mode] + ["SQL " + s for s in sqls]
    p = subprocess.run([PROBE], input="\n".join(cmds) + "\n", capture_output=True, text=True)
    if p.returncode != 0:
        raise SystemExit("probe failed in MODE %s: %s" % (mode, p.stderr[-2000:]))
    return [json.loads(l) for l in p.stdout.splitlines() if l.strip()]


def canon(res):
    def key(r):
        return [(0, "") if c is None else (1, repr(c)) for c in r]
    return sorted((list(r) for r in res["rows"]), key=key)


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    total = lowered = errors_equal = 0
    bad = []
    with tempfile.TemporaryDirectory() as tmp:
        for table, cs in cases(first, count).items():
            img, _, kind = refcases.table_image(table)
            path = os.path.join(tmp, table + ".cst")
            with open(path, "wb") as f:
                f.write(img)
            sqls = [c["sql"] for c in cs]
            cpu = run("cpu", path, kind, sqls)
            gpu = run("gpu", path, kind, sqls)
            assert len(cpu) == len(gpu) == len(cs)
            for c, a, b in zip(cs, cpu, gpu):
                total += 1
                d = [x for x in b.get("decisions", []) if x["node"] == "groupby"]
                lowered += 1 if d and d[0]["lowered"] else 0
                if not a["ok"] or not b["ok"]:
                    # an error is an error in both (division by zero, ...)
                    if a["ok"] != b["ok"] or ("zero" in a.get("error", "")) != ("zero" in b.get("error", "")):
                        bad.append((c["id"], c["sql"], a.get("error"), b.get("error")))
                    else:
                        errors_equal += 1
                    continue
                if a["types"] != b["types"] or canon(a) != canon(b):
                    bad.append((c["id"], c["sql"], len(a["rows"]), len(b["rows"])))
            print("[soak] %s: %d queries done" % (table, len(cs)), flush=True)
    print(json.dumps(dict(queries=total, lowered_to_gpu=lowered, both_failed_alike=errors_equal,
                          mismatches=len(bad))))
    for x in bad[:10]:
        print("MISMATCH", x)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
