"""where does a bench step spend its time outside the kernel? (config 3 shape)"""
import sys
import time

sys.path.insert(0, ".")
import torch  # noqa: F401  (loads the HIP runtime first)
import eventql_amd as E
from eventql_amd import bench_plans as B

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
ctx = E.Context(0)
t = ctx.generate(rows, "kabv")
q = t.query(B.config3())
for _ in range(3):
    q.launch(); q.finish()
    while q.next_batch(1024)[0]:
        pass
N = 30
tl = tf = td = 0.0
kms = 0.0
for _ in range(N):
    t0 = time.perf_counter()
    q.launch()
    t1 = time.perf_counter()
    q.finish()
    t2 = time.perf_counter()
    while q.next_batch(1024)[0]:
        pass
    t3 = time.perf_counter()
    tl += t1 - t0
    tf += t2 - t1
    td += t3 - t2
    kms += q.stats()["kernel_ms"]
print("launch %.3f ms  finish %.3f ms (kernel %.3f)  drain %.3f ms" %
      (tl / N * 1e3, tf / N * 1e3, kms / N, td / N * 1e3))
