set -o pipefail
run() { name=$1; shift; env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline $FLAGS > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err; python - <<PY
import json
d=json.loads(open("gpurun_out/ab_$name.json").read().strip().splitlines()[-1])
print("$name", d["value"], "it ms", d["roofline"]["iteration"]["ms_graph"])
PY
}
FLAGS="--mlp-fp16" run f16_motion_first IMMOCO_FORK2=motion
FLAGS="--mlp-fp16" run f16_image_first IMMOCO_FORK2=image
FLAGS="" run f32_image_first IMMOCO_FORK2=image
