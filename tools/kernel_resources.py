#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage for one .hip file:
prints kernels with scratch (spills) or VGPR count above a threshold.
usage: tools/kernel_resources.py pygpukit_amd/csrc/engine.hip [min_vgpr]"""
import re, subprocess, sys, os
src = sys.argv[1]
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 129
root = os.path.dirname(os.path.abspath(__file__)) + "/.."
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{root}/include",
       f"-I{os.path.dirname(src)}", "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
rows = []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line) or re.search(r" Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    for key, pat in (("vgpr", r"VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"), ("sgpr", r"SGPRs: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
n = 0
for r in rows:
    if r.get("scratch", 0) > 0 or r.get("vgpr", 0) >= thr:
        n += 1
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()[:110]
        print(f"vgpr={r.get('vgpr')} scratch={r.get('scratch')} occ={r.get('occ')} lds={r.get('lds')} {name}")
print(f"{len(rows)} kernels, {n} listed (scratch>0 or vgpr>={thr})")
