#!/usr/bin/env python3
"""Soak on an MI355X box: the device-side cstable writer (evql_table_from_device_columns)
against the host writer (byte-identical to the reference's CSTableWriter,
tests/test_gpu_writer.py) on random row counts, column subsets / orders and NULL rates:
the images must be equal byte for byte.  usage: tests/soak_writer.py <first seed> <count>"""
import json
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import eventql_amd as E  # noqa: E402
import test_gpu_writer as W  # noqa: E402

SIZES = [0, 1, 2, 63, 64, 65, 127, 128, 129, 1023, 1024, 1025, 2047, 2048, 2049, 65535, 65536, 65537,
         131071, 131072, 131073, 262145]


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    ctx = E.Context(0)
    bad = []
    for seed in range(first, first + count):
        r = random.Random(seed)
        n = r.choice(SIZES) if r.random() < 0.6 else r.randrange(0, 400_000)
        c = W.make_columns(n, seed)
        rate = r.choice([0.0, 0.01, 0.3, 0.5, 0.99, 1.0])
        rng = np.random.default_rng(seed)
        for name in list(c):
            if name.endswith("_null"):
                c[name] = (rng.random(n) < rate).astype(np.uint8)
        pool = W.REQUIRED + W.OPTIONAL
        specs = r.sample(pool, r.randint(1, len(pool)))
        t = W.device_table(ctx, specs, c, n)
        dev = t.download_image()
        t.close()
        host = W.host_image(specs, c, n)
        if dev != host:
            bad.append((seed, n, rate, [s["name"] for s in specs], len(dev), len(host)))
        if (seed - first) % 20 == 19:
            print("[writer soak] %d done" % (seed - first + 1), flush=True)
    print(json.dumps(dict(tables=count, mismatches=len(bad))))
    for b in bad[:10]:
        print("MISMATCH", b)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
