#!/usr/bin/env python3
"""Per-kernel register / scratch metadata of the gfx950 code object inside kueue_amd/libkq_engine.so
(llvm-readelf --notes: .vgpr_count, .sgpr_count, spill counts, .private_segment_fixed_size, LDS)."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kueue_amd", "libkq_engine.so")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
txt = ""
with tempfile.TemporaryDirectory() as t:
    subprocess.check_call([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, f"{t}/fat.bin"])
    fat = open(f"{t}/fat.bin", "rb").read()
    # one bundle per translation unit (kq_engine.hip, kq_spec_kernel.hip), concatenated in the section
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), fat)]
    for i, a in enumerate(starts):
        b = starts[i + 1] if i + 1 < len(starts) else len(fat)
        open(f"{t}/b{i}.bin", "wb").write(fat[a:b])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={t}/b{i}.bin",
                               f"--output={t}/kq{i}.co", "--unbundle"])
        txt += subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", f"{t}/kq{i}.co"], text=True)
for blk in txt.split("- .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
    name = re.sub(r"^_Z\d+", "", g("name"))[:26]
    print(f"{name:26s} vgpr {g('vgpr_count'):>4s} agpr {blk.split()[0]:>3s} sgpr {g('sgpr_count'):>4s} vgpr_spill {g('vgpr_spill_count'):>4s} "
          f"sgpr_spill {g('sgpr_spill_count'):>4s} scratch {g('private_segment_fixed_size'):>6s} B  lds {g('group_segment_fixed_size'):>6s} B")
