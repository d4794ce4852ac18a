#!/usr/bin/env python3
"""Soak on an MI355X box: random GROUP BYs over the LSM partitions of tests/lsm_tables.py
through the REFERENCE's engine -- its own PartitionCursor + CPU operators (MODE cpu) against
the GPU operator found through the adapter's registry / resolver (MODE gpu:
gpu_partition.h -> evql_query_create_chain) -- FINAL rows and PartialGroupBy rows.
usage: tests/soak_partition_differential.py <first seed> <count per partition>"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import lsm_tables  # noqa: E402
import refcases  # noqa: E402
from refcases import RefGen, _case, LSM  # noqa: E402

PROBE = os.path.join(ROOT, "oracle", "_ref", "csql_probe")


def canon(rows):
    return sorted((list(r) for r in rows), key=lambda row: [(0, "") if v is None else (1, repr(v)) for v in row])


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    total = lowered = alike = 0
    bad = []
    with tempfile.TemporaryDirectory() as tmp:
        for pi, pname in enumerate(sorted(lsm_tables.PARTITIONS)):
            d = os.path.join(tmp, pname)
            os.mkdir(d)
            specs = []
            for fname, img, skl, upd, _ in refcases.partition_files("lsm:" + pname):
                with open(os.path.join(d, fname + ".cst"), "wb") as f:
                    f.write(img)
                specs.append("%s:%d:%d" % (fname, skl, upd))
            cs = []
            for seed in range(first, first + count):
                g = RefGen(seed * 16 + pi, **LSM)
                g.count_cols = ["k", "a", "n", "v"]
                c = _case("lsm-%s-s%d" % (pname, seed), "lsm:" + pname, g.plan_kwargs([1]),
                          lsm_tables.LSM_SCHEMA)
                if c:
                    cs.append(c)
            out = {}
            for mode in ("cpu", "gpu"):
                cmds = ["PARTITION t %s %s" % (d, " ".join(specs)), "ROWS on", "MODE " + mode]
                cmds += ["SQL " + c["sql"] for c in cs]
                cmds += ["MODE " + mode + " partial"] + ["SQL " + c["sql"] for c in cs]
                p = subprocess.run([PROBE], input="\n".join(cmds) + "\n", capture_output=True, text=True)
                if p.returncode != 0:
                    raise SystemExit("probe failed in MODE %s on %s: %s" % (mode, pname, p.stderr[-2000:]))
                out[mode] = [json.loads(l) for l in p.stdout.splitlines() if l.strip()]
            for i, (a, b) in enumerate(zip(out["cpu"], out["gpu"])):
                c = cs[i % len(cs)]
                part = i >= len(cs)
                total += 1
                dd = [x for x in b.get("decisions", []) if x["node"] == "groupby"]
                if part and not (dd and dd[0]["lowered"]):
                    continue
                lowered += 1 if dd and dd[0]["lowered"] else 0
                if not a["ok"] or not b["ok"]:
                    if a["ok"] != b["ok"] or ("zero" in a.get("error", "")) != ("zero" in b.get("error", "")):
                        bad.append((c["id"], part, c["sql"][:200], a.get("error"), b.get("error")))
                    else:
                        alike += 1
                    continue
                if a["types"] != b["types"] or canon(a["rows"]) != canon(b["rows"]):
                    bad.append((c["id"], part, c["sql"][:200], len(a["rows"]), len(b["rows"])))
            print("[partition soak] %s: %d queries x 2 modes" % (pname, len(cs)), flush=True)
    print(json.dumps(dict(results=total, lowered_to_gpu=lowered, both_failed_alike=alike, mismatches=len(bad))))
    for x in bad[:10]:
        print("MISMATCH", x)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
