#!/usr/bin/env python3
"""Instruction mix and register counts of one kernel in a gfx950 code object kept by the build
(`ZK_KEEP_HSACO=build/hsaco make -C zksnake_amd/csrc ...`).  usage: isa_stats.py <file.hsaco> <kernel name substring>"""
import collections
import re
import subprocess
import sys

LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    path, want = sys.argv[1], sys.argv[2]
    dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", path], capture_output=True, text=True, check=True).stdout
    meta = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", path], capture_output=True, text=True).stdout
    blocks = re.split(r"\n(?=[0-9a-f]+ <)", dis)
    for blk in blocks:
        head = blk.split("\n", 1)[0]
        m = re.match(r"[0-9a-f]+ <(.+)>:", head)
        if not m or want not in m.group(1) or m.group(1).endswith(".kd"):
            continue
        name = m.group(1)
        ops = collections.Counter()
        for line in blk.split("\n")[1:]:
            parts = line.split()
            if parts and re.match(r"^[a-z_0-9]+$", parts[0]) and (parts[0].startswith(("v_", "s_", "ds_", "global_", "buffer_", "flat_", "scratch_"))):
                ops[parts[0]] += 1
        total = sum(ops.values())
        valu = sum(c for o, c in ops.items() if o.startswith("v_"))
        mads = ops.get("v_mad_u64_u32", 0)
        print(f"{name}\n  instructions {total}  VALU {valu}  v_mad_u64_u32 {mads}  v_mul_lo_u32 {ops.get('v_mul_lo_u32', 0)}  "
              f"ds {sum(c for o, c in ops.items() if o.startswith('ds_'))}  global {sum(c for o, c in ops.items() if o.startswith('global_'))}  "
              f"scratch {sum(c for o, c in ops.items() if o.startswith('scratch_'))}  s_waitcnt {ops.get('s_waitcnt', 0)}  s_nop {ops.get('s_nop', 0)}")
        print("  top:", ", ".join(f"{o} {c}" for o, c in ops.most_common(14)))
        # register counts from the kernel's metadata record (.name, .vgpr_count, .vgpr_spill_count appear in this order per kernel)
        rec = re.search(r"\.name:\s+" + re.escape(name) + r"\s(?:.|\n)*?\.vgpr_count:\s+(\d+)\s*\n\s*\.vgpr_spill_count:\s+(\d+)", meta)
        if rec:
            print(f"  vgpr_count {rec.group(1)}  vgpr_spill_count {rec.group(2)}")


if __name__ == "__main__":
    main()
