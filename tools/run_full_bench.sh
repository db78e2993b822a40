mkdir -p gpurun_out
timeout -k 10 900 python bench.py > gpurun_out/bench_full.log 2>&1; echo exit=$?
tail -1 gpurun_out/bench_full.log
