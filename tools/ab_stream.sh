# entry-stream ablation: isolated motion_encode_bwd time with the full stream vs half of it (timing only)
for mode in 3 7; do
for flags in "" "--mlp-fp16"; do
IMMOCO_CSR_STREAM=$mode python bench.py --steps 1 --warmup 1 --iters 300 --no-cpu-baseline $flags > gpurun_out/ab_stream.json 2> gpurun_out/ab_stream.err
python - <<PY
import json
d=json.loads(open("gpurun_out/ab_stream.json").read().strip().splitlines()[-1])
print("IMMOCO_CSR_STREAM=$mode flags='$flags' motion_encode_bwd isolated ms", d["roofline"]["kernels_ms_isolated"]["motion_encode_bwd"], "graph iteration ms", d["roofline"]["iteration"]["ms_graph"])
PY
done; done
