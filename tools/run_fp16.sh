mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -x > gpurun_out/t6.log 2>&1; echo exit=$? >> gpurun_out/t6.log; tail -5 gpurun_out/t6.log
for f in "" "--table-fp16"; do
timeout -k 10 200 python bench.py --iters 300 --steps 2 --warmup 1 --no-cpu-baseline $f > gpurun_out/b6$f.log 2>&1 && tail -1 gpurun_out/b6$f.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms']; print(d['dtype'], d['value'], d['psnr_db'], d['roofline']['iteration']['ms_graph'], {n: k[n] for n in ('motion_encode_fwd','image_encode_fwd','adam_motion','adam_image','motion_encode_bwd')})"
done
