#!/usr/bin/env python3
"""Prints the bench lines of profiles/<tag>_bench_*.json as a markdown table.
usage: bench_table.py <tag>"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
print("| file | workload | rate | ms/step | kernel ms | algorithmic GB/s | frac | traffic GB | drain |")
print("|---|---|---|---|---|---|---|---|---|")
for path in sorted(glob.glob(os.path.join(ROOT, "profiles", tag + "_bench_*.json"))):
    for line in open(path):
        if not line.startswith("{"):
            continue
        d = json.loads(line)
        r = d.get("roofline") or {}
        c = d.get("config", {})
        drain = c.get("drain_after_last_step") or {}
        tr = r.get("traffic")
        print("| %s | %s | %.3g %s | %.3f | %s | %s | %s | %s | %s |" % (
            os.path.basename(path), c.get("workload"), d["value"], d["unit"], d["ms_per_step"],
            "%.3f" % r["kernel_ms"] if r.get("kernel_ms") else "-",
            "%.0f" % r["achieved"] if r.get("achieved") else "-",
            "%.3f" % r["frac"] if r.get("frac") else "-",
            "%.2f" % (tr / 1e9) if tr else "-",
            json.dumps(drain) if drain else "-"))
        cb = d.get("cpu_baseline")
        if cb:
            print("|  | cpu_baseline | %.3g %s (%s, %s cores) | | | | | | %s |" % (
                cb["value"], cb["unit"], cb.get("kind"), cb.get("cores"), cb.get("sample", "")[:80]))
