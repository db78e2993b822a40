set -o pipefail
python -m pytest tests -m gpu -x -q -k "first_steps or returns_last or fp16_tables or non_square or single_group or mlp_fp16_96" > gpurun_out/ab_tests.log 2>&1; tail -3 gpurun_out/ab_tests.log
run() { name=$1; shift; env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline $FLAGS > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err; python - <<PY
import json
d=json.loads(open("gpurun_out/ab_$name.json").read().strip().splitlines()[-1])
print("$name", d["value"], "it ms", d["roofline"]["iteration"]["ms_graph"], d["psnr_db"]["solved"])
PY
}
FLAGS="" run f32_k1 IMMOCO_GRAPH_K=1
FLAGS="" run f32_k8 IMMOCO_GRAPH_K=8
FLAGS="" run f32_k8_early IMMOCO_GRAPH_K=8 IMMOCO_FORK=early
FLAGS="--mlp-fp16" run f16_k1_late IMMOCO_GRAPH_K=1 IMMOCO_FORK=late
FLAGS="--mlp-fp16" run f16_k8_late IMMOCO_GRAPH_K=8 IMMOCO_FORK=late
FLAGS="--mlp-fp16" run f16_k8_early IMMOCO_GRAPH_K=8 IMMOCO_FORK=early
FLAGS="--mlp-fp16 --table-fp16" run f16t_k8_early IMMOCO_GRAPH_K=8 IMMOCO_FORK=early
