mkdir -p gpurun_out
for early in 0 1 0 1; do
if [ $early = 1 ]; then export IMMOCO_FORK_EARLY=1; else unset IMMOCO_FORK_EARLY; fi
timeout -k 10 200 python bench.py --iters 300 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/b12_$early.log 2>&1 && tail -1 gpurun_out/b12_$early.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('early=$early', d['value'], d['roofline']['iteration']['ms_graph'], d['roofline']['kernel_ms'])"
done
