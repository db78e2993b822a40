set -o pipefail
run() { name=$1; shift; env "$@" python bench.py --workload c3 --batch 4 --steps 1 --warmup 1 --iters 600 --no-cpu-baseline --pair --mlp-fp16 > gpurun_out/pair_$name.json 2> gpurun_out/pair_$name.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/pair_$name.json").read().strip().splitlines()[-1])
    print("$name", "slices/s", d["value"], "ms per slice-iteration", d["roofline"]["iteration"]["ms_graph"])
except Exception as e:
    print("$name", "ERR", e); print(open("gpurun_out/pair_$name.err").read()[-1500:])
PY
}
run pad12k_default IMMOCO_X=1
run pad30k IMMOCO_CSR_PAD_LDS=30000
run pad0 IMMOCO_CSR_PAD_LDS=0
run pad30k_k1 IMMOCO_CSR_PAD_LDS=30000 IMMOCO_GRAPH_K=1
