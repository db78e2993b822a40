mkdir -p gpurun_out/prof2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -k "solver or adam or plan" > gpurun_out/t12.log 2>&1; echo exit=$? >> gpurun_out/t12.log; tail -3 gpurun_out/t12.log
timeout -k 10 300 python bench.py --iters 300 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/b11.log 2>&1; echo exit=$?
tail -1 gpurun_out/b11.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['psnr_db'], d['roofline']['iteration']); print(d['roofline']['kernels_ms'])"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof2/kt -- python bench.py --iters 100 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/prof2/kt.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/prof2/pmc1 -- python bench.py --iters 20 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/prof2/pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d gpurun_out/prof2/pmc2 -- python bench.py --iters 20 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/prof2/pmc2.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum --kernel-trace --output-format csv -d gpurun_out/prof2/pmc3 -- python bench.py --iters 20 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/prof2/pmc3.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof2/pmc4 -- python bench.py --iters 20 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/prof2/pmc4.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof2/pmc5 -- python bench.py --iters 20 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/prof2/pmc5.log 2>&1
ls gpurun_out/prof2/*/runc/ | head -30
