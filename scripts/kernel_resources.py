#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel of libzkcnn_hip.so, read from the code-object metadata.

Compiles each .hip translation unit device-only for gfx950 (same flags as zkcnn_amd/build.py) and prints one markdown row per kernel:
VGPRs, AGPRs, SGPRs, spills, private segment (scratch bytes per lane), LDS bytes, waves per SIMD the register count allows
(512 VGPRs per SIMD lane, allocation granule 8).   usage: scripts/kernel_resources.py [name-regex] > profiles/rNN_kernel_resources.md"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "zkcnn_amd", "csrc")
flt = sys.argv[1] if len(sys.argv) > 1 else ""
print("| kernel | vgpr | agpr | sgpr | vgpr spills | scratch B/lane | LDS B | waves/SIMD |")
print("|---|---|---|---|---|---|---|---|")
with tempfile.TemporaryDirectory() as tmp:
    for src in ("hip/sumcheck.hip", "hip/witness.hip", "hip/verifier.hip", "hip/hyrax.hip"):
        co = os.path.join(tmp, "x.co")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "--cuda-device-only", "--no-gpu-bundle-output", "-O3", "-std=c++17",
                               "-Wno-unused-value", "-I" + CSRC, "-c", os.path.join(CSRC, src), "-o", co])
        txt = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co], text=True)
        for blk in txt.split("  - .agpr_count:")[1:]:
            blk = ".agpr_count:" + blk

            def g(k):
                m = re.search(r"\." + k + r":\s+(\S+)", blk)
                return m.group(1) if m else "?"
            name = g("name")
            if flt and not re.search(flt, name):
                continue
            tot = max(int(g("vgpr_count")) + int(g("agpr_count")), 1)
            waves = min(8, 512 // (((tot + 7) // 8) * 8))
            name = re.sub(r"^_Z\d+", "", name)
            print(f"| {name} | {g('vgpr_count')} | {g('agpr_count')} | {g('sgpr_count')} | {g('vgpr_spill_count')} | {g('private_segment_fixed_size')} | "
                  f"{g('group_segment_fixed_size')} | {waves} |")
