#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X.

A step = one pass of the hot path over one synthetic cstable partition that is
already resident in HBM: fused scan -> filter -> GROUP BY kernel, (N > 1) exchange
and merge of the partial aggregates across ranks, compaction of the group table
and delivery of every result row through `nextBatch` (packed SVector bytes).

Workloads (config.workload):
  config3 (default)  BASELINE.json configs[2], the configuration the metric is
           quoted on: 1e9 rows, 4 columns (k, a, b uint64 PLAIN, v float64),
           `WHERE a > 30000 AND b < 30000`, `k, sum(v), count(1), sum(b) GROUP BY k`,
           1000 groups.  N GPUs: every rank scans its own 1e9-row partition
           (EventQL partitions shard onto GPUs, weak scaling); partial aggregates
           travel as dense group records: RCCL all_gather + merge kernel.
  config2  `k, sum(v), count(1) GROUP BY k`, 2 columns (16 B/row).
  config4  high cardinality: key u uniform in [0, 1e7), 3 aggregates; records are
           hash-partitioned by identity % N and exchanged with RCCL all_to_all,
           every rank merges the key range it owns.

usage: python bench.py [--gpus N] [--steps K] [--warmup W] [--rows R] [--workload ...]
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

QUERIES = {
    "config2": ("k, sum(v), count(1) GROUP BY k", "kv", 2),
    "config3": ("k, sum(v), count(1), sum(b) WHERE a>30000 AND b<30000 GROUP BY k", "kabv", 4),
    # config 3 over the reference's default integer encoding (k, a, b UINT64_LEB128)
    "config3l": ("k, sum(v), count(1), sum(b) WHERE a>30000 AND b<30000 GROUP BY k", "kabv", 4),
    "config4": ("u, sum(a), count(1), sum(v) GROUP BY u  (u uniform in [0,1e7))", "uav", 3),
    # BASELINE configs[3] as written: the key is a STRING ("g" + u, STRING_PLAIN)
    "config4s": ("s, sum(a), count(1), sum(v) GROUP BY s  (s = 'g' + u, u uniform in [0,1e7), "
                 "STRING_PLAIN key)", None, 3),
    # nested: REPEATED RECORD items{position, price}, Dremel flattening (CSTableScan)
    "config5": ("items.position, count(1), sum(items.price) GROUP BY items.position "
                "(REPEATED RECORD items, rlevel 1 / dlevel 2)", None, 2),
    # record scan (CSTableScan, AGGREGATE_WITHIN_RECORD_FLAT) under the GROUP BY
    "config5w": ("n, count(1), sum(s) GROUP BY n  over  (count(items.position) WITHIN RECORD "
                 "AS n, sum(items.price) WITHIN RECORD AS s)", None, 2),
}


def config5_plan():
    from eventql_amd import capi as K
    from eventql_amd.plan import Plan, col, count, sum_
    S = {"id": K.T_UINT64, "items.position": K.T_UINT64, "items.price": K.T_UINT64,
         "score": K.T_FLOAT64}
    return Plan(S, select=[col("items.position"), count(1), sum_(col("items.price"))],
                group_by=[col("items.position")], scan_mode=K.SCAN_NESTED, groups_hint=16)


def config5w_plan():
    from eventql_amd import capi as K
    from eventql_amd.plan import Plan, col, count, sum_, out
    S = {"id": K.T_UINT64, "items.position": K.T_UINT64, "items.price": K.T_UINT64,
         "score": K.T_FLOAT64}
    return Plan(S, scan_select=[count(col("items.position")), sum_(col("items.price"))],
                select=[out(0), count(1), sum_(out(1))], group_by=[out(0)],
                scan_mode=K.SCAN_NESTED_WITHIN_RECORD, groups_hint=16)


REFERENCE_SQL = {
    "config3": "select k, sum(a), count(1), sum(b) from t where a > 30000 and b < 30000 group by k;",
    "config4": "select u, sum(a), count(1) from t where a >= 0 group by u;",
}


def cpu_baseline(ctx, plan_fn, columns, sample_rows, workload=None, **gen_kw):
    """the oracle (CPU restatement of the reference path, 1 thread) on a bounded
    sample of the same workload"""
    import oracle_lib as O
    t = ctx.generate(sample_rows, columns, **gen_kw)
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
    out = dict(value=sample_rows / dt, unit="rows/s", cores=1, kind="port",
               sample="%d-row instance of the same table/query, oracle (C restatement of "
                      "FastCSTableScan+VM+GroupBy), %d groups, %.1f s"
                      % (sample_rows, res.nrows, dt))
    # the reference runs one thread per partition (one partition per connection):
    # the same sample as T concurrent partitions on the host's cores, for scale
    T = max(1, min(16, len(os.sched_getaffinity(0))))
    if T > 1:
        dtp = O.oracle_time_parallel(path, [plan_fn() for _ in range(T)])
        out["all_cores"] = dict(value=T * sample_rows / dtp, unit="rows/s", cores=T,
                                sample="%d threads, each the same %d-row partition, %.1f s"
                                       % (T, sample_rows, dtp))
    # the REFERENCE's own engine where its build travelled here (oracle/_ref/csql_probe:
    # csql::Runtime, parser, planner, VM, GroupByExpression over FastCSTableScan, compiled
    # from the reference's sources by oracle/ref_csql/build.sh): timed on the same file.
    # This snapshot of the reference has no sum(float64): its run uses the integer twin
    # of the query (sum(a) for sum(v)).
    probe = os.path.join(ROOT, "oracle", "_ref", "csql_probe")
    ref_sql = REFERENCE_SQL.get(workload)
    if os.path.exists(probe) and ref_sql:
        import subprocess
        cmds = "TABLE t %s fast\nROWS off\nTIME 1 %s\n" % (path, ref_sql)
        try:
            p = subprocess.run([probe], input=cmds, capture_output=True, text=True, timeout=600)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
            r = json.loads(line)
            if r.get("ok"):
                port = dict(out)
                out = dict(value=sample_rows / r["seconds"], unit="rows/s", cores=1, kind="reference",
                           sample="%d-row instance of the same table, the reference's own csql "
                                  "engine (1 thread, as one partition runs): %s -- %d groups, %.1f s"
                                  % (sample_rows, ref_sql, r["nrows"], r["seconds"]),
                           port=port)
        except Exception as e:  # the baseline is a reported figure, never a reason to fail
            out["reference_error"] = str(e)[:200]
    os.unlink(path)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=0, help="rows per GPU (0 = workload default)")
    ap.add_argument("--workload", default="config3", choices=sorted(QUERIES))
    ap.add_argument("--cpu-sample-rows", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --rows per GPU; strong: --rows in total, split over the GPUs")
    ap.add_argument("--no-hint", action="store_true",
                    help="plans without groups_hint (the reference's planner has none): the "
                         "first execute estimates the cardinality itself")
    ap.add_argument("--float-sums", default="fast", choices=["fast", "exact"],
                    help="exact: EVQL_FLOAT_SUM_EXACT (order-independent, bit-stable float sums)")
    ap.add_argument("--k-bits", type=int, default=0,
                    help="config2 run B (SURVEY.md 8d): k as UINT32_BITPACKED of this width")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a
    # version banner on stdout when a communicator is created): from here on file
    # descriptor 1 is stderr, and the result line goes to the saved descriptor.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import eventql_amd as E
    from eventql_amd import bench_plans as B, capi as K, distributed as D, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # EVQL_DIST_BACKEND=gloo rehearses the N > 1 path with several ranks on ONE
    # GPU (records staged through the host); the real runs use RCCL ("nccl")
    backend = os.environ.get("EVQL_DIST_BACKEND", "nccl")
    device = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend)

    query_text, columns, ncols = QUERIES[args.workload]
    high_card = args.workload in ("config4", "config4s")
    string_keys = args.workload == "config4s"
    n_keys = 10_000_000
    nested = args.workload in ("config5", "config5w")
    # config 4 = 1.25e8 rows per partition, config 5 = 1e8 records
    rows = args.rows or (125_000_000 if high_card else
                         (100_000_000 if nested or args.workload == "config3l" else 1_000_000_000))
    if args.scaling == "strong":
        rows = rows // world  # the job's rows are fixed, every rank scans its share
    fs = dict(float_sum_mode=K.FLOAT_SUM_EXACT) if args.float_sums == "exact" else {}
    plan_fn = {"config2": lambda **kw: B.config2(**fs, **kw),
               "config3": lambda **kw: B.config3(**fs, **kw),
               "config3l": lambda **kw: B.config3(**fs, **kw),
               "config5": config5_plan, "config5w": config5w_plan,
               "config4": lambda **kw: B.config4(groups_hint=0 if args.no_hint else n_keys, **kw),
               "config4s": lambda **kw: B.config4s(groups_hint=0 if args.no_hint else n_keys, **kw)
               }[args.workload]
    gen_kw = dict(u_mod=n_keys) if high_card else {}
    if args.k_bits:
        gen_kw["k_bits"] = args.k_bits

    ctx = E.Context(device)
    # every rank owns one partition; different seeds => different partitions
    seed = synth.SEED if rank == 0 else (synth.SEED + 0x9E3779B97F4A7C15 * rank) & synth.MASK
    nested_image = None
    leb = args.workload == "config3l"
    materialize_ms = None
    if leb:
        # config 3's table with the reference's DEFAULT integer encoding
        # (UINT64_LEB128, TableSchema.cc:290-316), written by the host writer; the
        # first operator re-encodes the three columns on the device, once, as
        # bit-packed pages of the narrowest of 8 / 16 / 32 bits
        from eventql_amd import capi as K
        c = synth.table_columns(rows, seed=seed)
        w = E.Writer([dict(name=n, logical_type=K.COL_UNSIGNED_INT,
                           storage_type=K.ENC_UINT64_LEB128) for n in "kab"] +
                     [dict(name="v", logical_type=K.COL_FLOAT, storage_type=K.ENC_FLOAT_IEEE754)])
        for n in "kabv":
            w.put(n, c[n])
        w.commit(rows)
        table = ctx.open_image(w.image())
        w.close()
        del c
        ctx.synchronize()
        t0m = time.perf_counter()
        table.query(plan_fn()).close()
        ctx.synchronize()
        materialize_ms = (time.perf_counter() - t0m) * 1e3
    elif string_keys:
        # generated in HBM and encoded by the device writer; the first operator finds
        # the value boundaries of the STRING_PLAIN stream and hashes the keys, once
        table = B.string_key_table(ctx, rows, n_keys, seed)
        ctx.synchronize()
        t0m = time.perf_counter()
        table.query(plan_fn()).close()
        ctx.synchronize()
        materialize_ms = (time.perf_counter() - t0m) * 1e3
    elif nested:
        # `rows` = records; written by the host writer (levels + LEB128 / bit-packed
        # data), then resident in HBM like any other table
        nested_image, nested_stats = synth.items_table_image(rows, seed=3 + rank)
        table = ctx.open_image(nested_image)
    else:
        table = ctx.generate(rows, columns, seed=seed, **gen_kw)
    ctx.synchronize()
    t0q = time.perf_counter()
    q = table.query(plan_fn())
    ctx.synchronize()
    first_operator_ms = (time.perf_counter() - t0q) * 1e3
    # N > 1: the exchange step runs behind the C ABI (evql_query_exchange): RCCL
    # send/recv between the GPUs, python only hands the ncclUniqueId around
    xchg = D.make_exchange(ctx) if world > 1 else None
    xmode = K.EXCHANGE_BY_OWNER if high_card else K.EXCHANGE_GATHER_ALL

    def drain(qq):
        n = 0
        while True:
            k, _ = qq.next_batch(1024)
            if k == 0:
                return n
            n += k

    per_query = args.workload == "config5w"

    def step():
        if per_query and world == 1:
            # the per-record reduction belongs to the operator, not to the table: a
            # step builds the operator (kernel from the cache), reduces the records,
            # runs the GROUP BY over them and drains it
            qq = table.query(plan_fn())
            qq.launch()
            qq.finish()
            n = drain(qq)
            step.stats = qq.stats()
            # both kernels of the operator: k_within_record (per-record reduction) and
            # evql_scan_agg (the GROUP BY over the per-record values)
            step.kernel_ms = step.stats["total_ms"]
            qq.close()
            return n
        q.launch()
        q.finish()
        if world == 1:
            # 1e7 result rows are not pulled through nextBatch inside the timed
            # region (the reference puts ORDER BY / LIMIT above such a GROUP BY)
            return q.stats()["num_groups"] if high_card else drain(q)
        # partial aggregates -> records bucketed by owner (high cardinality) or sent to
        # everybody (low) -> merged in rank order; low: rank 0's result is the answer,
        # high: the result stays distributed over the ranks
        q.exchange(xchg, xmode)
        return q.stats()["num_groups"] if high_card else drain(q)

    for _ in range(args.warmup):
        step()
    kernel_ms = []
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ngroups_out = step()
        kernel_ms.append(getattr(step, "kernel_ms", None) or q.stats()["kernel_ms"])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        rdev = "cuda" if backend == "nccl" else "cpu"
        tmax = torch.tensor([dt], dtype=torch.float64, device=rdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        g = torch.tensor([ngroups_out if high_card else 0], dtype=torch.int64, device=rdev)
        dist.all_reduce(g)
        if high_card:
            ngroups_out = int(g.item())

    stats = getattr(step, "stats", None) or q.stats()
    drain_info = None
    if high_card and world == 1:
        # OUTSIDE ms_per_step: every group of the last step pulled through nextBatch
        # (GroupByExpression::nextBatch -> packed SVector bytes, 1024 rows per call as the
        # reference's ResultCursor asks)
        torch.cuda.synchronize()
        t0d = time.perf_counter()
        nd = drain(q)
        ddt = time.perf_counter() - t0d
        drain_info = dict(groups=int(nd), drain_ms=ddt * 1e3, groups_per_s=nd / ddt,
                          batch_rows=1024)
    if rank == 0:
        total_rows = rows * world * args.steps
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
                    traffic = json.load(f).get("%s_%d" % (args.workload, rows), {}).get(
                        "hbm_bytes_per_launch")
            except Exception:
                traffic = None
        # the kernels `kernel_ms` covers (hipEvents around them on the context's stream)
        if args.workload == "config5w":
            kernel_names = "k_within_record + evql_scan_agg"
        elif "evql_part_scatter" in q.kernel_source():
            src = q.kernel_source()
            kernel_names = " + ".join(k for k in ("evql_part_count", "evql_part_scatter",
                                                  "evql_part_refine", "evql_part_aggregate")
                                      if (k + "(") in src)
        else:
            kernel_names = "evql_scan_agg"
        merge = "none"
        if world > 1:
            merge = ("%s: group records bucketed by owner on the device, all-to-all, merged in "
                     "rank order" if high_card else
                     "%s: every rank's group records to every rank, merged in rank order"
                     ) % xchg.backend()
            out_x = xchg.stats()
        out = {
            # BASELINE.json's metric, quoted on config3; the other workloads say what they are
            "metric": ("rows/sec scanned+aggregated, 1e9-row 4-col GROUP BY"
                       if args.workload == "config3" else
                       "rows/sec scanned+aggregated, %s (%d rows/GPU)" % (args.workload, rows)),
            "value": total_rows / dt,
            "unit": "rows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "u64/f64",
            "data": "synthetic",
            "config": {
                "workload": args.workload,
                "query": query_text,
                "rows_per_gpu": rows,
                "columns": ncols,
                "encodings": ("k, a, b UINT64_LEB128 (re-encoded once per table as 16-bit bit-packed pages), v FLOAT_IEEE754"
                              if leb else "UINT64_PLAIN/FLOAT_IEEE754" if not args.k_bits else
                              "k UINT32_BITPACKED(%d bit), others UINT64_PLAIN/FLOAT_IEEE754"
                              % args.k_bits),
                "groups": int(ngroups_out),
                "partitions": world,
                "merge": merge,
                "groups_hint": "none (estimated by the first execute)" if args.no_hint else "given",
                "float_sums": args.float_sums,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": kernel_names,
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": alg_bytes,
                "kernel_ms": avg_kernel_ms,
            },
        }
        if world == 1 and not args.no_cpu_baseline and not leb:
            if string_keys:
                import oracle_lib as O
                n_s = args.cpu_sample_rows or 4_000_000
                ts = B.string_key_table(ctx, n_s, n_keys, seed)
                img_s = ts.download_image()
                ts.close()
                t0c = time.time()
                res = O.oracle_run(img_s, plan_fn())
                dtc = time.time() - t0c
                out["cpu_baseline"] = dict(
                    value=n_s / dtc, unit="rows/s", cores=1, kind="port",
                    sample="%d-row instance of the same table/query, oracle (C restatement of "
                           "FastCSTableScan+VM+GroupBy, SHA1-keyed map), %d groups, %.1f s"
                           % (n_s, res.nrows, dtc))
            elif nested:
                import oracle_lib as O
                n_s = min(rows, args.cpu_sample_rows or 2_000_000)
                img_s, _ = synth.items_table_image(n_s, seed=3)
                t0c = time.time()
                res = O.oracle_run(img_s, plan_fn())
                dtc = time.time() - t0c
                out["cpu_baseline"] = dict(
                    value=n_s / dtc, unit="records/s", cores=1, kind="port",
                    sample="%d-record instance, oracle (C restatement of CSTableScan "
                           "%s + GroupBy), %d groups, %.1f s" % (
                               n_s, "AGGREGATE_WITHIN_RECORD_FLAT" if args.workload == "config5w"
                               else "NO_AGGREGATION", res.nrows, dtc))
            else:
                sample = args.cpu_sample_rows or (4_000_000 if high_card else 80_000_000)
                out["cpu_baseline"] = cpu_baseline(ctx, plan_fn, columns, sample,
                                                   workload=args.workload, **gen_kw)
        if drain_info:
            out["config"]["drain_after_last_step"] = drain_info
        if world > 1:
            # phases of the exchange step of the last timed step on rank 0 (evql_exchange_
            # last_stats: export = records + first rows + owner buckets, transfer = the
            # all_gather of counts + all-to-all, merge = merge kernels + recount), beside
            # the scan kernels' device time
            out_x["scan_kernel_ms"] = avg_kernel_ms
            out["config"]["exchange_last_step"] = out_x
        if leb or string_keys:
            out["config"]["materialize_ms_first_operator"] = materialize_ms
        if string_keys:
            out["config"]["encodings"] = ("s STRING_PLAIN (value boundaries, 64-bit hashes and the exact "
                                          "dictionary of dense 32-bit codes built on the device once per "
                                          "table), a UINT64_PLAIN, v FLOAT_IEEE754")
        if nested:
            # the Dremel flattening (level decode, slot maps, LEB128 decode) runs when
            # the first operator over these columns is created (together with the
            # hiprtc compile of the plan) and is cached on the table; later operators
            # reuse the flattened columns
            ctx.synchronize()
            t0p = time.perf_counter()
            q2 = table.query(plan_fn())
            ctx.synchronize()
            out["config"]["first_operator_ms_flatten_and_jit"] = first_operator_ms
            out["config"]["next_operator_ms"] = (time.perf_counter() - t0p) * 1e3
            q2.close()
            out["config"]["records_per_gpu"] = rows
            out["config"]["flattened_rows_per_gpu"] = int(stats["rows_scanned"])
            out["config"]["encodings"] = "levels bit-packed, position UINT32_BITPACKED(4b), price LEB128"
            out["unit"] = "records/s"
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
