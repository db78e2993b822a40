#!/usr/bin/env python
"""Per-launch averages of the L2-request PMC pass of tools/final_profile.sh (its own rocprofv3 --pmc run):

    python tools/pmc_l2_requests.py gpurun_out/final r02_final    ->  profiles/r02_final_l2_requests.csv
"""
import csv, glob, os, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, tag = sys.argv[1], sys.argv[2]
COUNTERS = ["TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCC_HIT_sum", "TCC_MISS_sum"]
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
files = glob.glob(os.path.join(src, "l2req", "*", "*_counter_collection.csv"))
assert files, f"no counter_collection.csv under {src}/l2req"
for f in files:
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        if row["Counter_Name"] not in COUNTERS:
            continue
        name = name.split("(")[0].replace("void ", "") + f" [grid {row['Grid_Size']}]"
        a = acc[name][row["Counter_Name"]]
        a[0] += 1
        a[1] += float(row["Counter_Value"])
out = os.path.join(ROOT, "profiles", f"{tag}_l2_requests.csv")
with open(out, "w") as fh:
    fh.write("kernel,launches," + ",".join(COUNTERS) + ",note: per-launch averages; rocprofv3 --pmc (own pass) of "
             "python bench.py --iters 20 --steps 1 --warmup 0 --no-cpu-baseline\n")
    for name, d in sorted(acc.items(), key=lambda kv: -kv[1][COUNTERS[0]][1] / max(kv[1][COUNTERS[0]][0], 1)):
        n = d[COUNTERS[0]][0]
        fh.write(f'"{name}",{n},' + ",".join(f"{d[c][1] / max(d[c][0], 1):.0f}" for c in COUNTERS) + "\n")
print("".join(open(out).readlines()[:8]))
