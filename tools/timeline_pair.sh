set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/tl
rm -rf gpurun_out/tl/pair
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl/pair -- python3 bench.py --workload c3 --batch 2 --pair --mlp-fp16 --iters 48 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/tl/pair.log 2>&1
python3 - <<'PY' > gpurun_out/tl/pair.txt
import csv, glob, re
f = glob.glob("gpurun_out/tl/pair/*/*_kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
def short(n): return re.sub(r"^void ", "", n).replace("immoco::", "").split("(")[0][:44]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows)
ticks = [i for i, e in enumerate(ev) if e[2].startswith("tick_kernel")]
a, b = ticks[40], ticks[44]
t0 = ev[a][1]
print(f"{'start us':>9} {'end us':>9} {'dur us':>8}  kernel   (two double iterations of the paired graph)")
for s, e, n in ev[a + 1:b + 1]:
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {n}")
PY
rm -rf gpurun_out/tl/pair
cat gpurun_out/tl/pair.txt
