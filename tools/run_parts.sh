mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -k "plan or solver" > gpurun_out/t5.log 2>&1; echo exit=$? >> gpurun_out/t5.log; tail -3 gpurun_out/t5.log
for gp in 1 2 4 8; do
timeout -k 10 200 python bench.py --iters 200 --steps 1 --warmup 1 --no-cpu-baseline --grad-parts $gp > gpurun_out/b4_$gp.log 2>&1; echo parts=$gp exit=$?
tail -1 gpurun_out/b4_$gp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms']; print(d['value'], d['psnr_db']['solved'], d['roofline']['iteration']['ms_graph'], 'enc_bwd', k['motion_encode_bwd'], 'adam_mot', k['adam_motion'])"
done
