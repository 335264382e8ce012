#!/usr/bin/env python3
"""Registers, scratch, LDS and occupancy of every kernel of rt_kernels.hip as the compiler reports them
(-Rpass-analysis=kernel-resource-usage with the Makefile's flags), plus the spill traffic in the ISA of the tree kernels.
    python tools/resource_usage.py [extra compiler flags ...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "raytracer_amd", "csrc", "rt_kernels.hip")
flags = ["-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC"] + sys.argv[1:]
p = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950"] + flags + ["-c", src, "-o", "/tmp/ru.o", "-Rpass-analysis=kernel-resource-usage"],
                   capture_output=True, text=True)
cur, rows = None, []
for line in p.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line) or re.search(r"remark: .*?Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}; rows.append(cur); continue
    for key, pat in (("sgpr", r"TotalSGPRs: (\d+)"), ("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("sgpr_spill", r"SGPRs Spill: (\d+)"), ("vgpr_spill", r"VGPRs Spill: (\d+)"),
                     ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
print("%-70s %5s %5s %5s %8s %4s %6s %6s %7s" % ("kernel", "SGPR", "VGPR", "AGPR", "scratch", "occ", "sSpill", "vSpill", "LDS"))
for r in rows:
    if "rt_" in r["name"]:
        print("%-70s %5s %5s %5s %8s %4s %6s %6s %7s" % (r["name"][:70], r.get("sgpr"), r.get("vgpr"), r.get("agpr"), r.get("scratch"), r.get("occ"),
                                                     r.get("sgpr_spill"), r.get("vgpr_spill"), r.get("lds")))
