set -x
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 -L > gpurun_out/prof/counters.txt 2>&1 || true
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/kt -- python bench.py --iters 100 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/prof/kt.log 2>&1
echo kt exit=$?
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d gpurun_out/prof/pmc1 -- python bench.py --iters 20 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/prof/pmc1.log 2>&1
echo pmc1 exit=$?
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum --kernel-trace --output-format csv -d gpurun_out/prof/pmc2 -- python bench.py --iters 20 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/prof/pmc2.log 2>&1
echo pmc2 exit=$?
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof/pmc3 -- python bench.py --iters 20 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/prof/pmc3.log 2>&1
echo pmc3 exit=$?
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof/pmc4 -- python bench.py --iters 20 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/prof/pmc4.log 2>&1
echo pmc4 exit=$?
find gpurun_out/prof -name "*.csv" | head -30
du -sh gpurun_out/prof
