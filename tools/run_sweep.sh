mkdir -p gpurun_out
for mode in sweep tile; do for f in "" "--table-fp16"; do
IMMOCO_ENCODE_FWD=$mode timeout -k 10 200 python bench.py --iters 300 --steps 2 --warmup 1 --no-cpu-baseline $f > gpurun_out/b7_$mode$f.log 2>&1 && tail -1 gpurun_out/b7_$mode$f.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_isolated']; print('$mode', d['dtype'], d['value'], d['psnr_db']['solved'], d['roofline']['iteration']['ms_graph'], {n: k[n] for n in ('motion_encode_fwd','image_encode_fwd')})"
done; done
timeout -k 10 500 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -x > gpurun_out/t8.log 2>&1; echo exit=$? >> gpurun_out/t8.log; tail -3 gpurun_out/t8.log
