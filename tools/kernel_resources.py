#!/usr/bin/env python3
"""Registers / spills / occupancy of the kernels in one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py annonet_amd/csrc/kernels_mfma.hip [name filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Iinclude", "-Rpass-analysis=kernel-resource-usage",
       "-c", src, "-o", "/dev/null"] + sys.argv[3:]
run = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
t = run.stderr
names = re.findall(r"Function Name: ([A-Za-z0-9_]+)", t)
if run.returncode != 0 or not names:   # (c++filt with no arguments would wait on stdin)
    sys.exit("\n".join(l for l in t.splitlines() if "error" in l) or t[-2000:])
dem = dict(zip(names, subprocess.run(["c++filt"] + names, capture_output=True, text=True, stdin=subprocess.DEVNULL).stdout.splitlines()))
for b in re.split(r"remark: [^\n]*Function Name: ", t)[1:]:
    name = b.split("\n")[0].strip()
    d = re.sub(r"anh::\(anonymous namespace\)::", "", dem.get(name, name))
    d = re.sub(r"\(anh::.*$", "", d)
    if flt and flt not in d:
        continue
    g = lambda k: re.search(k + r": (\d+)", b).group(1)
    print(d[:86].ljust(88), "VGPR", g("VGPRs").rjust(3), "AGPR", g("AGPRs").rjust(3), "spill", g("VGPRs Spill").rjust(3), "scratch", g(r"ScratchSize \[bytes/lane\]").rjust(4),
          "occ", g(r"Occupancy \[waves/SIMD\]"), "LDS", g(r"LDS Size \[bytes/block\]"))
