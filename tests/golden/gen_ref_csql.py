#!/usr/bin/env python3
"""Generates tests/golden/ref_csql_<suite>.json by running the cases of
tests/refcases.py through the REAL reference csql engine.

Needs /root/reference (this container only): oracle/ref_csql/build.sh compiles the
reference's sources in place, with its own vendored protoc for the *.pb.h the csql
headers need, and links oracle/_ref/csql_probe.  The probe executes every SQL text
with the reference's unmodified CPU operators (GroupByExpression over FastCSTableScan /
CSTableScan; PartialGroupByExpression built as server/sql/scheduler.cc:79-115 does)
and dumps the vm::Programs its compiler produced.  Only the resulting JSON travels.

    python tests/golden/gen_ref_csql.py [suite ...]
"""
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PROBE = os.path.join(ROOT, "oracle", "_ref", "csql_probe")
MAX_ROWS_INLINE = 32


def run_probe(commands):
    p = subprocess.run([PROBE], input="\n".join(commands) + "\n", capture_output=True,
                       text=True, check=True)
    return [json.loads(l) for l in p.stdout.splitlines() if l.strip()]


def canon_probe_rows(types, rows):
    """probe cells -> the canonical form of sqlgen.canon_rows"""
    import sqlgen
    from eventql_amd import capi as K
    out = []
    for r in rows:
        rr = []
        for t, c in zip(types, r):
            if c is None:
                rr.append(None)
            elif t == K.T_FLOAT64:
                rr.append(c)  # already "f:<bits>"
            else:
                rr.append(c)
        out.append(rr)

    def key(r):
        return [(0, "") if c is None else (1, repr(c)) for c in r]
    out.sort(key=key)
    return out


def dremel_scan_well_defined(programs):
    cols, sel = programs["scan_columns"], programs["scan_select"]
    if len(cols) != len(sel):
        return False
    return all(p.get("lowerable") and [c[:3] for c in p["code"]] == [[4, cols[i][1], i], [7, 0, 0]]
               for i, p in enumerate(sel))


def pack_result(res):
    import sqlgen
    if not res["ok"]:
        return dict(ok=False, error=res["error"])
    rows = canon_probe_rows(res["types"], res["rows"])
    d = dict(ok=True, types=res["types"], nrows=res["nrows"], digest=sqlgen.rows_digest(rows))
    d["rows"] = rows if len(rows) <= MAX_ROWS_INLINE else rows[:MAX_ROWS_INLINE]
    return d


def pack_partial(res):
    """PartialGroupByExpression rows (groupby.cc:438-472): (20-byte SHA1 key, state
    bytes) as hex pairs, sorted by key"""
    import sqlgen
    if not res["ok"]:
        return dict(ok=False, error=res["error"])
    pairs = sorted([r[0].encode("latin-1").hex(), r[1].encode("latin-1").hex()]
                   for r in res["rows"])
    d = dict(ok=True, nrows=res["nrows"], digest=sqlgen.rows_digest(pairs))
    # inline sample: [key, first 128 state bytes, state length]; the digest covers all
    d["sample"] = [[k, v[:256], len(v) // 2] for k, v in pairs[:16]]
    return d


def main(argv):
    import refcases
    if not os.path.exists(PROBE):
        subprocess.check_call([os.path.join(ROOT, "oracle", "ref_csql", "build.sh")])
    want = argv or list(refcases.SUITES)
    tmp = tempfile.mkdtemp(prefix="refcsql")
    files = {}
    for name in want:
        cases = refcases.SUITES[name]()
        by_table = {}
        for c in cases:
            by_table.setdefault(c["table"], []).append(c)
        out_cases = []
        for table, cs in by_table.items():
            if table.startswith("lsm:"):
                # a partition: the reference's own PartitionCursor over a PartitionSnapshot
                # with these lsm_tables (probe.cc PARTITION)
                if table not in files:
                    specs = []
                    for fname, img, skl, upd, _ in refcases.partition_files(table):
                        open(os.path.join(tmp, fname + ".cst"), "wb").write(img)
                        specs.append("%s:%d:%d" % (fname, skl, upd))
                    files[table] = ("PARTITION t %s %s" % (tmp, " ".join(specs)), "fast")
                table_cmd, kind = files[table]
            else:
                if table not in files:
                    img, _, kind = refcases.table_image(table)
                    path = os.path.join(tmp, table + ".cst")
                    open(path, "wb").write(img)
                    files[table] = ("TABLE t %s %s" % (path, kind), kind)
                table_cmd, kind = files[table]
            cmds = [table_cmd, "DUMP on", "ROWS on", "MODE cpu"]
            cmds += ["SQL " + c["sql"] for c in cs]
            cmds += ["DUMP off", "MODE cpu partial"]
            cmds += ["SQL " + c["sql"] for c in cs]
            res = run_probe(cmds)
            assert len(res) == 2 * len(cs), (len(res), len(cs))
            for i, c in enumerate(cs):
                final, part = res[i], res[len(cs) + i]
                assert final["sql"] == c["sql"]
                oc = dict(id=c["id"], table=c["table"], sql=c["sql"], result=pack_result(final))
                if "programs" in final:
                    oc["programs"] = final["programs"]
                    if kind == "dremel" and not dremel_scan_well_defined(final["programs"]):
                        # CSTableScan declares its columns in input order and fills them
                        # in select-list order (sqlgen.make_runnable): values of the wrong
                        # width are popped off the VM stack.  Kept for the bytecode
                        # comparison; the rows are not a contract
                        oc["result"] = dict(ok=None, undefined="CSTableScan column order")
                # a global aggregate has no group key: the reference's partial
                # operator still emits one row; keep it
                if "select" in c["kw"]:
                    oc["partial"] = pack_partial(part)
                else:  # a bare scan has no partial form
                    oc["partial"] = dict(ok=False, error="bare scan")
                out_cases.append(oc)
        out_cases.sort(key=lambda c: c["id"])
        doc = dict(
            _source="generated by tests/golden/gen_ref_csql.py: the reference's own csql "
                    "engine (csql::Runtime; sources compiled in place by "
                    "oracle/ref_csql/build.sh) run over tables built by tests/refcases.py",
            suite=name, cases=out_cases)
        dst = os.path.join(HERE, "ref_csql_%s.json" % name)
        with open(dst, "w") as f:
            json.dump(doc, f, separators=(",", ":"))
            f.write("\n")
        nok = sum(1 for c in out_cases if c["result"]["ok"])
        print("%s: %d cases (%d ok, %d reference errors) -> %s (%d KiB)" % (
            name, len(out_cases), nok, len(out_cases) - nok, dst, os.path.getsize(dst) // 1024))


if __name__ == "__main__":
    main(sys.argv[1:])
