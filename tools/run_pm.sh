mkdir -p gpurun_out
for pm in 0 1; do
if [ $pm = 1 ]; then export IMMOCO_CSR_PARTMAJOR=1; fi
timeout -k 10 200 python bench.py --iters 300 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/b9_$pm.log 2>&1 && tail -1 gpurun_out/b9_$pm.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_isolated']; print('partmajor=$pm', d['value'], d['roofline']['iteration']['ms_graph'], d['roofline']['kernel_ms'], {n: k[n] for n in ('motion_encode_bwd','adam_motion')})"
done
