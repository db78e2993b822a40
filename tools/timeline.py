#!/usr/bin/env python
"""Per-kernel start/end of ONE replayed iteration from a rocprofv3 kernel trace
(`rocprofv3 --kernel-trace --output-format csv -d DIR -- python bench.py --iters 40 --steps 1 --warmup 0
--no-cpu-baseline`): shows what runs beside what in the forked graph.
    python tools/timeline.py DIR/<host>/<pid>_kernel_trace.csv [iteration_index]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
k = int(sys.argv[2]) if len(sys.argv) > 2 else 20


def short(n):
    return re.sub(r"^void ", "", n).replace("immoco::", "").split("(")[0][:60]


ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows)
ticks = [i for i, e in enumerate(ev) if e[2].startswith("tick_kernel")]
a, b = ticks[k], ticks[k + 1]
t0 = ev[a][1]
print(f"{'start us':>9} {'end us':>9} {'dur us':>8}  kernel")
for s, e, n in ev[a + 1:b + 1]:
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {n}")
