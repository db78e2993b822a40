set -o pipefail
python -m pytest tests -m gpu -x -q -k "first_steps or returns_last or fp16_tables or non_square or single_group or config1 or batch_pair or teacher_forced_state" > gpurun_out/chain_tests.log 2>&1; tail -3 gpurun_out/chain_tests.log
run() { name=$1; shift; env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline $FLAGS > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err; python - <<PY
import json
d=json.loads(open("gpurun_out/ab_$name.json").read().strip().splitlines()[-1])
print("$name", d["value"], "it ms", d["roofline"]["iteration"]["ms_graph"], d["psnr_db"]["solved"])
PY
}
FLAGS="" run f32_join IMMOCO_CHAIN_ITERS=0
FLAGS="" run f32_chain IMMOCO_CHAIN_ITERS=1
FLAGS="--mlp-fp16" run f16_join IMMOCO_CHAIN_ITERS=0
FLAGS="--mlp-fp16" run f16_chain IMMOCO_CHAIN_ITERS=1
FLAGS="--mlp-fp16" run f16_chain_k16 IMMOCO_CHAIN_ITERS=1 IMMOCO_GRAPH_K=16
