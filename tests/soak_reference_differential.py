#!/usr/bin/env python3
"""Differential soak on an MI355X box: random GROUP BY queries through the REFERENCE's own
engine (oracle/_ref/csql_probe, built from /root/reference by oracle/ref_csql/build.sh),
once with its CPU operators (MODE cpu) and once with the GPU operator plugged in (MODE gpu);
rows must be identical.  Same generators as the committed fixtures (tests/refcases.py),
other seeds; flat tables (incl. string-producing select expressions) and the two nested
ones (Dremel scans).
usage: tests/soak_reference_differential.py <first seed offset> <count> [families, e.g. items,testtbl
       | all] [partial]"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))  # (test infrastructure: lives under tests/)
import refcases  # noqa: E402
import sqlgen  # noqa: E402
import tables as T  # noqa: E402
from refcases import RefGen, RefNestedGen, RefStrGen, MIXED, MIXED_COUNT_COLS, ITEMS, TESTTBL, _case  # noqa: E402
from eventql_amd import capi as K  # noqa: E402

PROBE = os.path.join(ROOT, "oracle", "_ref", "csql_probe")


def dremel_scan_well_defined(programs):
    """CSTableScan declares its columns in input order and fills them in select-list
    order: outside this regime the reference pops values of the wrong width off its VM
    stack (tests/golden/gen_ref_csql.py) and its rows are not a contract"""
    cols, sel = programs["scan_columns"], programs["scan_select"]
    if len(cols) != len(sel):
        return False
    return all(p.get("lowerable") and [c[:3] for c in p["code"]] == [[4, cols[i][1], i], [7, 0, 0]]
               for i, p in enumerate(sel))


def cases(first, count):
    import nested_tables as N
    out = {"mixed": [], "ranges": [], "items": [], "testtbl": [], "strings": []}
    for seed in range(first, first + count):
        which = seed % 2
        table, schema, cols = (("items", N.ITEMS_SCHEMA, ITEMS) if which == 0 else
                               ("testtbl", N.NESTED_SCHEMA, TESTTBL))
        g = RefNestedGen(170_000 + seed, **cols)
        g.schema = schema
        g.count_cols = list(cols["uint_cols"]) + list(cols["bool_cols"])
        if which == 0:
            g.leaf_uint, g.leaf_bool = ["items.position", "items.price"], []
        else:
            g.leaf_uint = ["event.search_query.result_items.position"]
            g.leaf_bool = ["event.search_query.result_items.clicked"]
        c = _case("nested-s%d" % seed, table, g.plan_kwargs([1]), schema, scan_mode=K.SCAN_NESTED)
        if c:
            out[table].append(c)
        g = RefGen(150_000 + seed, **MIXED)
        g.count_cols = MIXED_COUNT_COLS
        c = _case("mixed-s%d" % seed, "mixed", g.plan_kwargs([1]), T.MIXED_SCHEMA)
        if c:
            out["mixed"].append(c)
        g = RefStrGen(180_000 + seed, **MIXED)
        g.count_cols = MIXED_COUNT_COLS
        c = _case("strings-s%d" % seed, "mixed", g.plan_kwargs([1]), T.MIXED_SCHEMA)
        if c:
            out["strings"].append(c)
        g = RefGen(160_000 + seed, **T.RANGES)
        c = _case("ranges-s%d" % seed, "ranges", g.plan_kwargs([1]), T.RANGES_SCHEMA)
        if c:
            out["ranges"].append(c)
    return out


PARTIAL = len(sys.argv) > 4 and sys.argv[4] == "partial"


def run(mode, path, kind, sqls):
    # partial: PartialGroupByExpression's rows (group key, saved states) as bytes
    cmds = ["TABLE t %s %s" % (path, kind), "ROWS on", "MODE " + mode + (" partial" if PARTIAL else "")]
    if mode == "cpu" and kind == "dremel":
        cmds.append("DUMP on")
    cmds += ["SQL " + s for s in sqls]
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
    total = lowered = errors_equal = undefined = 0
    bad = []
    with tempfile.TemporaryDirectory() as tmp:
        want = sys.argv[3].split(",") if len(sys.argv) > 3 and sys.argv[3] != "all" else None
        for table, cs in cases(first, count).items():
            if want and table not in want:
                continue
            img, _, kind = refcases.table_image("mixed" if table == "strings" else table)
            path = os.path.join(tmp, table + ".cst")
            with open(path, "wb") as f:
                f.write(img)
            sqls = [c["sql"] for c in cs]
            cpu = run("cpu", path, kind, sqls)
            gpu = run("gpu", path, kind, sqls)
            assert len(cpu) == len(gpu) == len(cs)
            for c, a, b in zip(cs, cpu, gpu):
                if kind == "dremel" and ("programs" not in a or
                                         not dremel_scan_well_defined(a["programs"])):
                    undefined += 1
                    continue
                total += 1
                d = [x for x in b.get("decisions", []) if x["node"] == "groupby"]
                if PARTIAL and not (d and d[0]["lowered"]):
                    continue  # (the probe's CPU fallback in gpu mode is the final operator)
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
                          skipped_reference_undefined=undefined, mismatches=len(bad))))
    for x in bad[:10]:
        print("MISMATCH", x)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
