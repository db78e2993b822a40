# round-end evidence: full default bench + rocprofv3 kernel stats of the same command (short iters)
mkdir -p gpurun_out/final
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err; echo bench exit=$?
tail -1 gpurun_out/final/bench_default.json | cut -c1-600
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/kt -- python bench.py --iters 200 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/final/kt.log 2>&1; echo kt exit=$?
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/final/fetch -- python bench.py --iters 20 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/final/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/final/write -- python bench.py --iters 20 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/final/write.log 2>&1
