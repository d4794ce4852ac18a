#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X.

A step = one pass of the hot path (fused scan -> filter -> GROUP BY kernel,
partial-aggregate merge across ranks, group-table compaction and fetch) over
one synthetic cstable partition that is already resident in HBM.

Workload (config.workload = "config3"): BASELINE.json configs[2], the
configuration the metric is quoted on -- 1e9 rows, 4 columns (k, a, b uint64
PLAIN, v float64), `WHERE a > 30000 AND b < 30000`, `k, sum(v), count(1),
sum(b) GROUP BY k`, 1000 groups.  With N GPUs every rank scans its own
1e9-row partition (EventQL partitions shard onto GPUs; weak scaling) and the
per-rank partial aggregates are merged with an RCCL all_gather of the dense
group records followed by a merge kernel on every rank.

usage: python bench.py [--gpus N] [--steps K] [--warmup W] [--rows R]
       (N > 1: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(ctx, plan_fn, sample_rows):
    """the oracle (CPU restatement of the reference path, 1 thread) on a bounded
    sample of the same workload"""
    import oracle_lib as O
    t = ctx.generate(sample_rows, "kabv")
    img = t.download_image()
    t.close()
    path = "/tmp/evql_bench_sample.cst"
    with open(path, "wb") as f:
        f.write(img)
    del img
    plan = plan_fn()
    t0 = time.time()
    res = O.oracle_run(path, plan)
    dt = time.time() - t0
    os.unlink(path)
    return dict(value=sample_rows / dt, unit="rows/s", cores=1, kind="port",
                sample="%d-row prefix-shaped instance of the same table/query, "
                       "oracle (C restatement of FastCSTableScan+VM+GroupBy), %d groups, %.1f s"
                       % (sample_rows, res.nrows, dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=1_000_000_000, help="rows per GPU")
    ap.add_argument("--workload", default="config3", choices=["config2", "config3"])
    ap.add_argument("--cpu-sample-rows", type=int, default=40_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import eventql_amd as E
    from eventql_amd import bench_plans as B

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    plan_fn = B.config3 if args.workload == "config3" else B.config2
    ctx = E.Context(local_rank)
    # every rank owns one partition; different seeds => different partitions
    from eventql_amd import synth
    table = ctx.generate(args.rows, "kabv", seed=synth.SEED + 0x9E3779B97F4A7C15 * rank
                         if rank else synth.SEED)
    ctx.synchronize()
    q = table.query(plan_fn())
    rw = q.record_words()
    max_groups = 4096
    if world > 1:
        send = torch.zeros(max_groups * rw + 1, dtype=torch.int64, device="cuda")
        recv = torch.zeros(world * (max_groups * rw + 1), dtype=torch.int64, device="cuda")

    def step():
        q.launch()
        q.finish()
        if world > 1:
            # partial aggregates -> dense records -> all ranks -> merge kernel
            n = q.export_groups(send.data_ptr() + 8, max_groups)
            send[0] = n
            dist.all_gather_into_tensor(recv, send)
            torch.cuda.synchronize()
            counts = recv.view(world, -1)[:, 0].tolist()
            for r in range(world):
                if r == rank or counts[r] == 0:
                    continue
                q.import_groups(recv.view(world, -1)[r, 1:].data_ptr(), counts[r])

    for _ in range(args.warmup):
        step()
    kernel_ms = []
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        kernel_ms.append(q.stats()["kernel_ms"])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    stats = q.stats()
    result = q.fetch_all()
    if rank == 0:
        total_rows = args.rows * world * args.steps
        avg_kernel_ms = sum(kernel_ms) / len(kernel_ms)
        # per-launch algorithmic bytes: payload of the referenced
        # column streams + result bytes
        alg_bytes = stats["algorithmic_bytes"]
        achieved = alg_bytes / (avg_kernel_ms * 1e-3) / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                with open(tf) as f:
                    tj = json.load(f)
                key = "%s_%d" % (args.workload, args.rows)
                traffic = tj.get(key, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "rows/sec scanned+aggregated, 1e9-row 4-col GROUP BY",
            "value": total_rows / dt,
            "unit": "rows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64/f64",
            "data": "synthetic",
            "config": {
                "workload": args.workload,
                "query": "k, sum(v), count(1), sum(b) WHERE a>30000 AND b<30000 GROUP BY k"
                         if args.workload == "config3" else "k, sum(v), count(1) GROUP BY k",
                "rows_per_gpu": args.rows,
                "columns": 4 if args.workload == "config3" else 2,
                "encodings": "UINT64_PLAIN/FLOAT_IEEE754",
                "groups": int(stats["num_groups"]),
                "partitions": world,
                "merge": "rccl all_gather of dense group records + merge kernel" if world > 1 else "none",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "evql_scan_agg",
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": alg_bytes,
                "kernel_ms": avg_kernel_ms,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(ctx, plan_fn, args.cpu_sample_rows)
        assert result.nrows == stats["num_groups"]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
