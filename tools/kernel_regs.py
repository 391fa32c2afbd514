#!/usr/bin/env python3
"""Register / LDS / scratch budget of every kernel in csrc/libnsc_hip.so, read from the code-object metadata
(llvm-objdump --offloading + llvm-readelf --notes).  `python tools/kernel_regs.py [substring]`."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_table(so=None):
    so = so or os.path.join(ROOT, "neural-spectral-codec_amd", "csrc", "libnsc_hip.so")
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        local = shutil.copy(so, os.path.join(tmp, "lib.so"))
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], check=True, capture_output=True, cwd=tmp)
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", os.path.join(tmp, f)], check=True,
                                   capture_output=True, text=True).stdout
            for blk in notes.split(".agpr_count")[1:]:
                def g(key):
                    m = re.search(rf"\.{key}:\s+(\S+)", blk)
                    return m.group(1) if m else None
                if g("name"):
                    out[g("name")] = dict(vgpr=int(g("vgpr_count")), sgpr=int(g("sgpr_count")),
                                          agpr=int(re.match(r":\s+(\d+)", blk).group(1)),
                                          lds=int(g("group_segment_fixed_size")),
                                          scratch=int(g("private_segment_fixed_size")))
    return out


if __name__ == "__main__":
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    demangle = shutil.which("c++filt") or f"{LLVM}/llvm-cxxfilt"
    for name, v in sorted(kernel_table().items()):
        if pat not in name:
            continue
        try:
            pretty = subprocess.run([demangle, name], capture_output=True, text=True).stdout.strip()[:110]
        except OSError:
            pretty = name
        print(f"{v['vgpr']:4d} vgpr {v['agpr']:3d} agpr {v['sgpr']:4d} sgpr {v['lds']:6d} lds {v['scratch']:5d} scratch  {pretty}")
