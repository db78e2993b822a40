set -o pipefail
python -m pytest tests -m gpu -x -q -k "mlp_fp16 or mlp_half or batch_pair or first_steps or fp16_tables" > gpurun_out/act16_tests.log 2>&1; tail -4 gpurun_out/act16_tests.log
run() { name=$1; shift; env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline $FLAGS > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err; python - <<PY
import json
d=json.loads(open("gpurun_out/ab_$name.json").read().strip().splitlines()[-1])
print("$name", d["value"], "it ms", d["roofline"]["iteration"]["ms_graph"], d["psnr_db"]["solved"])
print({k:v for k,v in d["roofline"]["kernels_ms_isolated"].items() if "motion" in k or "mlp" in k})
PY
}
FLAGS="--mlp-fp16" run f16_act16 X=1
FLAGS="--mlp-fp16 --grad-parts 2" run f16_act16_parts2 X=1
FLAGS="--mlp-fp16 --table-fp16" run f16t_act16 X=1
