#!/usr/bin/env python3
"""Static audit of generated kernels (no GPU needed): compiles plans for gfx950 through
evql_compile_only and reports, per kernel, VGPRs, spilled VGPRs, scratch, LDS bytes and
FLAT memory instructions (an LDS or global access whose address space the compiler could
not infer: it counts in vmcnt AND lgkmcnt and drains both, see evql_lds_peek).
usage: tests/isa_audit.py [number of fixture plans per suite, default 25]"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))  # (test infrastructure: lives under tests/)
import eventql_amd as E  # noqa: E402
from eventql_amd import bench_plans as B, capi as K  # noqa: E402
from eventql_amd.plan import Plan  # noqa: E402

LLVM = "/opt/rocm/lib/llvm/bin"


def audit(tag, plan, columns, seen):
    d = tempfile.mkdtemp()
    try:
        E.compile_only(plan, columns, d)
    except E.EvqlError as e:
        if e.code == K.EVQL_ENOTSUP:
            return
        raise
    for f in glob.glob(d + "/*.hsaco"):
        key = os.path.basename(f)
        if key in seen:
            continue
        seen.add(key)
        notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", f], capture_output=True, text=True).stdout
        asm = subprocess.run([LLVM + "/llvm-objdump", "-d", f], capture_output=True, text=True).stdout
        flat = {}
        cur = None
        for line in asm.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\w+)>:", line)
            if m:
                cur = m.group(1)
            elif "\tflat_" in line and cur:
                flat[cur] = flat.get(cur, 0) + 1
        for blk in notes.split("- .agpr_count")[1:]:
            name = re.search(r"\.name:\s+(\w+)", blk).group(1)
            g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
            row = (name, g("vgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"),
                   g("group_segment_fixed_size"), flat.get(name, 0))
            bad = row[2] or row[3] or row[5]
            print("%-28s %-22s vgpr %3d spill %3d scratch %4d lds %6d flat %d%s" %
                  ((tag[:28],) + row + ("   <==" if bad else "",)))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 25
    seen = set()
    narrow = [dict(name=c, logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT32_BITPACKED, bits=16)
              for c in "kab"] + [dict(name="v", logical_type=K.COL_FLOAT, storage_type=K.ENC_FLOAT_IEEE754)]
    audit("config2", B.config2(), B.PLAIN_COLUMNS, seen)
    audit("config3", B.config3(), B.PLAIN_COLUMNS, seen)
    audit("config3 16-bit pages", B.config3(), narrow, seen)
    audit("config4", B.config4(), B.PLAIN_COLUMNS, seen)
    audit("config4s", B.config4s(), B.STRING_KEY_COLUMNS, seen)
    import refcases
    import tables as T
    cols_of = {}
    for suite in ("mixed", "ranges"):
        img, schema, _ = refcases.table_image(suite)
        import oracle_lib as O
        path = tempfile.mktemp(suffix=".cst")
        with open(path, "wb") as f:
            f.write(img)
        rd = O.TableReader(path, "orc")
        info = rd.columns()
        rd.close()
        os.unlink(path)
        cols = [dict(name=c["name"], logical_type=c["logical_type"], storage_type=c["storage_type"],
                     dlevel_max=c["dlevel_max"], rlevel_max=c["rlevel_max"],
                     bits=(16 if c["storage_type"] in (K.ENC_UINT32_BITPACKED, K.ENC_BOOLEAN_BITPACKED) else 0))
                for c in info]
        for c in refcases.all_cases()[suite][:n]:
            try:
                plan = Plan(schema, **c["kw"])
            except Exception:
                continue
            audit(c["id"], plan, cols, seen)


if __name__ == "__main__":
    main()
