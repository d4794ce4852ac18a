"""debug aid: k_within_record time under EVQL_WR_DEBUG variants (1 = no atomics,
2 = no plain stores, 4 = no value loads)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eventql_amd as E
from eventql_amd import synth
sys.argv = sys.argv[:1]
import bench
ctx = E.Context(0)
img, _ = synth.items_table_image(int(os.environ.get("N", "100000000")), seed=3)
t = ctx.open_image(img)
t.query(bench.config5w_plan()).close()
for flags in (0, 1, 2, 3, 4, 7):
    os.environ["EVQL_WR_DEBUG"] = str(flags)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        t.query(bench.config5w_plan()).close()
    ctx.synchronize()
    print("flags", flags, "ms per operator create", (time.perf_counter() - t0) / 5 * 1e3, flush=True)
