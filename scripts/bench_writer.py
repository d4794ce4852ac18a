#!/usr/bin/env python3
"""Throughput of the device-side cstable writer (evql_table_from_device_columns):
N rows of SoA columns resident in HBM -> cstable v0.2.0 image in HBM.

usage: python scripts/bench_writer.py [rows]   (default 1e8)
Prints one JSON line per schema."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import eventql_amd as E  # noqa: E402
from eventql_amd import capi as K  # noqa: E402

U, F = K.COL_UNSIGNED_INT, K.COL_FLOAT


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
    ctx = E.Context(0)
    g = torch.Generator(device="cuda").manual_seed(1)
    wide = torch.randint(0, 1 << 62, (n,), dtype=torch.int64, device="cuda", generator=g)
    small = torch.randint(0, 1024, (n,), dtype=torch.int64, device="cuda", generator=g)
    mixed = wide >> torch.randint(0, 62, (n,), dtype=torch.int64, device="cuda", generator=g)
    nulls = (torch.rand((n,), device="cuda", generator=g) < 0.3).to(torch.uint8)
    torch.cuda.synchronize()
    schemas = {
        "config3 shape: 3 x UINT64_PLAIN + FLOAT_IEEE754": (
            [dict(name=x, logical_type=U, storage_type=K.ENC_UINT64_PLAIN) for x in "kab"] +
            [dict(name="v", logical_type=F, storage_type=K.ENC_FLOAT_IEEE754)],
            dict(k=small, a=wide, b=mixed, v=wide), {}),
        "UINT32_BITPACKED(10 bit) + UINT64_LEB128": (
            [dict(name="k", logical_type=U, storage_type=K.ENC_UINT32_BITPACKED,
                  bitpack_max_value=1023),
             dict(name="l", logical_type=U, storage_type=K.ENC_UINT64_LEB128)],
            dict(k=small, l=mixed), {}),
        "optional UINT64_PLAIN + optional UINT64_LEB128 (30% NULL)": (
            [dict(name="o", logical_type=U, storage_type=K.ENC_UINT64_PLAIN, dlevel_max=1),
             dict(name="ol", logical_type=U, storage_type=K.ENC_UINT64_LEB128, dlevel_max=1)],
            dict(o=wide, ol=mixed), dict(o=nulls, ol=nulls)),
    }
    for label, (specs, vals, nl) in schemas.items():
        v = {k: t.data_ptr() for k, t in vals.items()}
        m = {k: t.data_ptr() for k, t in nl.items()}
        best, size = None, 0
        for _ in range(3):
            ctx.synchronize()
            t0 = time.perf_counter()
            t = ctx.table_from_device_columns(specs, v, m, n)
            ctx.synchronize()
            dt = time.perf_counter() - t0
            size = int(E.lib().evql_table_image_size(t.h))
            t.close()
            best = dt if best is None else min(best, dt)
        in_bytes = n * (8 * len(vals) + len(nl))
        print(json.dumps({"schema": label, "rows": n, "ms": best * 1e3, "rows_per_s": n / best,
                          "input_GBps": in_bytes / best / 1e9, "file_bytes": size,
                          "file_GBps": size / best / 1e9}), flush=True)


if __name__ == "__main__":
    main()
