#!/usr/bin/env python
"""Registers / LDS / occupancy of every immoco kernel in a csrc file (hipcc -Rpass-analysis=kernel-resource-usage):
    python tools/kernel_resources.py csr mlp_mfma ..."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "miccai24_immoco_amd", "csrc")
for f in sys.argv[1:]:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                        "-c", os.path.join(CS, f + ".hip"), "-o", "/tmp/_kr.o", "-Rpass-analysis=kernel-resource-usage"],
                       capture_output=True, text=True)
    cur = {}
    for line in r.stderr.splitlines():
        m = re.search(r"remark: \S+ Function Name: (\S+)", line) or re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            continue
        for key in ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]", "TotalSGPRs"):
            m = re.search(re.escape(key) + r": (\d+)", line)
            if m and cur:
                cur[key] = int(m.group(1))
        if "LDS Size" in line and cur and "immoco" in cur["name"]:
            dem = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
            dem = re.sub(r"\(.*", "", dem).replace("immoco::", "").replace("void ", "")
            print(f"{f:10s} {dem[:60]:60s} vgpr {cur.get('VGPRs'):4d} agpr {cur.get('AGPRs'):4d} sgpr {cur.get('TotalSGPRs'):4d} "
                  f"scratch {cur.get('ScratchSize [bytes/lane]', 0):4d} occ {cur.get('Occupancy [waves/SIMD]')} lds {cur.get('LDS Size [bytes/block]')}")
            cur = {}
