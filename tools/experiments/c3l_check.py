import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import eventql_amd as E
from eventql_amd import bench_plans as B, capi as K, synth
rows = 100_000_000
ctx = E.Context(0)
c = synth.table_columns(rows, seed=synth.SEED)
w = E.Writer([dict(name=n, logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_LEB128) for n in "kab"] +
             [dict(name="v", logical_type=K.COL_FLOAT, storage_type=K.ENC_FLOAT_IEEE754)])
for n in "kabv":
    w.put(n, c[n])
w.commit(rows)
t = ctx.open_image(w.image())
w.close()
q = t.query(B.config3())
for _ in range(3):
    q.launch(); q.finish()
src = q.kernel_source()
print("unroll", re.search(r"#define EVQL_UNROLL (\d+)", src).group(1), "bitpacked", src.count("evql_bitpacked_x2<16>"),
      "kernel_ms", q.stats()["kernel_ms"])
