mkdir -p gpurun_out/traffic
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/traffic/fetch -- python bench.py --iters 20 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/traffic/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/traffic/write -- python bench.py --iters 20 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/traffic/write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/traffic/kt -- python bench.py --iters 200 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/traffic/kt.log 2>&1
echo done
