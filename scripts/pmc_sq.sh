#!/bin/bash
# SQ occupancy / stall counters of the evql_* kernels (one rocprofv3 --pmc pass per workload)
# usage: scripts/pmc_sq.sh <tag> <workload> [bench args]
set -eo pipefail
TAG=$1; WL=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --output-format csv -d "$OUT/sq_$WL" -- \
  python3 "$ROOT/bench.py" --workload "$WL" --steps 2 --warmup 1 --no-cpu-baseline "$@" > "$OUT/sq_$WL.log" 2>&1
python3 - "$OUT/sq_$WL" <<'PY'
import csv, glob, os, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("evql_"):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    wc = m.get("SQ_WAVE_CYCLES", 1)
    print(k, " ".join("%s=%.3g(%.0f%%)" % (n.replace("SQ_", ""), v, 100 * v / wc) for n, v in sorted(m.items())))
PY
