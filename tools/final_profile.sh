# round-end evidence: the default bench command, its rocprofv3 kernel stats, and the PMC traffic passes
# (separate --pmc runs, kernel trace only).  Every step must succeed before the next GPU step starts.
set -e -o pipefail
mkdir -p gpurun_out/final
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err
tail -1 gpurun_out/final/bench_default.json | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/kt_default -- python3 bench.py --no-cpu-baseline > gpurun_out/final/kt_default.log 2>&1
rm -f gpurun_out/final/kt_default/*/*_kernel_trace.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/final/fetch -- python3 bench.py --iters 20 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/final/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/final/write -- python3 bench.py --iters 20 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/final/write.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/final/l2req -- python3 bench.py --iters 20 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/final/l2req.log 2>&1
rm -f gpurun_out/final/fetch/*/*_kernel_trace.csv gpurun_out/final/write/*/*_kernel_trace.csv gpurun_out/final/l2req/*/*_kernel_trace.csv
ls gpurun_out/final/kt_default/*/
