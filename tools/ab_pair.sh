set -o pipefail
python -m pytest tests -m gpu -x -q -k "batch_pair or batch_of_slices or config3" > gpurun_out/pair_tests.log 2>&1; tail -5 gpurun_out/pair_tests.log
run() { name=$1; shift; python bench.py --workload c3 --batch 4 --steps 1 --warmup 1 --iters 600 --no-cpu-baseline "$@" > gpurun_out/pair_$name.json 2> gpurun_out/pair_$name.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/pair_$name.json").read().strip().splitlines()[-1])
    print("$name", "slices/s", d["value"], "ms per slice-iteration", d["roofline"]["iteration"]["ms_graph"], d["psnr_db"]["solved"])
except Exception as e:
    print("$name", "ERR", e); print(open("gpurun_out/pair_$name.err").read()[-1500:])
PY
}
run serial_f32
run pair_f32 --pair
run serial_f16 --mlp-fp16
run pair_f16 --pair --mlp-fp16
