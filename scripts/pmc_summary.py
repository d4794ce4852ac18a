#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc passes into profiles/traffic.json.

usage: pmc_summary.py <workload key> <fetch_dir> <write_dir> [kernel name prefix ...]

Per kernel (rows whose Kernel_Name starts with one of the prefixes; default: every
evql_* kernel) the average FETCH_SIZE / WRITE_SIZE per dispatch, corrected as
/opt/skills/guides/MI355X_MICROARCH.md section HBM prescribes for gfx950: FETCH_SIZE
reports half of the bytes of a wide coalesced streaming read (x2), WRITE_SIZE is exact
for 16-byte streaming stores; both are in KiB."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(directory, counter):
    out = {}
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != counter:
                    continue
                name = row["Kernel_Name"].split("(")[0]
                out.setdefault(name, []).append(float(row["Counter_Value"]))
    return out


def main():
    key, fdir, wdir = sys.argv[1:4]
    prefixes = sys.argv[4:] or ["evql_"]
    fetch, write = per_kernel(fdir, "FETCH_SIZE"), per_kernel(wdir, "WRITE_SIZE")
    kernels = {}
    total = 0.0
    for name in sorted(set(fetch) | set(write)):
        if not any(name.startswith(p) for p in prefixes):
            continue
        fv, wv = fetch.get(name, []), write.get(name, [])
        f_kib = sum(fv) / len(fv) if fv else 0.0
        w_kib = sum(wv) / len(wv) if wv else 0.0
        b = f_kib * 1024 * 2 + w_kib * 1024
        kernels[name] = dict(fetch_size_kib_raw=f_kib, write_size_kib_raw=w_kib,
                             hbm_bytes_per_launch=b, dispatches=max(len(fv), len(wv)))
        total += b
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    doc = json.load(open(tf)) if os.path.exists(tf) else {}
    doc[key] = dict(
        hbm_bytes_per_launch=total, kernels=kernels,
        correction="per kernel: FETCH_SIZE x1024 x2 (gfx950 reports half of wide coalesced "
                   "streaming reads) + WRITE_SIZE x1024; summed over the kernels of one step",
        source="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes "
               "(scripts/measure_round.sh)")
    with open(tf, "w") as f:
        json.dump(doc, f, indent=1)
        f.write("\n")
    print(json.dumps(doc[key], indent=1))


if __name__ == "__main__":
    main()
