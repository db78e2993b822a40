# round-4 evidence, part $1 (a | b | c): every GPU step must succeed before the next one starts.
set -e -o pipefail
mkdir -p gpurun_out/final
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ "$1" = "a" ]; then
  timeout -k 10 900 python3 bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err
  tail -n 1 gpurun_out/final/bench_default.json | cut -c1-300
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/kt_default -- python3 bench.py --no-cpu-baseline --no-alt-precision > gpurun_out/final/kt_default.log 2>&1
  rm -f gpurun_out/final/kt_default/*/*_kernel_trace.csv
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/final/fetch -- python3 bench.py --iters 20 --steps 1 --warmup 0 --no-cpu-baseline --no-alt-precision > gpurun_out/final/fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/final/write -- python3 bench.py --iters 20 --steps 1 --warmup 0 --no-cpu-baseline --no-alt-precision > gpurun_out/final/write.log 2>&1
  rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/final/l2req -- python3 bench.py --iters 20 --steps 1 --warmup 0 --no-cpu-baseline --no-alt-precision > gpurun_out/final/l2req.log 2>&1
  rm -f gpurun_out/final/fetch/*/*_kernel_trace.csv gpurun_out/final/write/*/*_kernel_trace.csv gpurun_out/final/l2req/*/*_kernel_trace.csv
  python3 bench.py --precision f16mlp --no-cpu-baseline --no-alt-precision > gpurun_out/final/bench_f16mlp.json 2> gpurun_out/final/bench_f16mlp.err
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/kt_f16mlp -- python3 bench.py --precision f16mlp --steps 1 --warmup 1 --no-cpu-baseline --no-alt-precision > gpurun_out/final/kt_f16mlp.log 2>&1
  rm -f gpurun_out/final/kt_f16mlp/*/*_kernel_trace.csv
  python3 bench.py --workload c5 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/final/bench_c5.json 2> gpurun_out/final/bench_c5.err
  mkdir -p gpurun_out/tl
  for mode in f32 f16mlp; do
    rm -rf gpurun_out/tl/$mode
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl/$mode -- python3 bench.py --iters 40 --steps 1 --warmup 0 --no-cpu-baseline --no-alt-precision --precision $mode > gpurun_out/tl/$mode.log 2>&1
    python3 tools/timeline.py gpurun_out/tl/$mode/*/*_kernel_trace.csv 20 > gpurun_out/final/timeline_$mode.txt
    rm -rf gpurun_out/tl/$mode
  done
  ls gpurun_out/final/kt_default/*/
elif [ "$1" = "c" ]; then   # the two pair lines alone, then the HIP side of the added 200-iteration cells (slices 2, 6, 7)
  python3 bench.py --workload c3 --batch 64 --steps 1 --warmup 0 --no-cpu-baseline --pair > gpurun_out/final/bench_c3_pair.json 2> gpurun_out/final/bench_c3_pair.err
  python3 bench.py --workload c3 --batch 64 --steps 1 --warmup 0 --no-cpu-baseline --pair --precision f16mlp > gpurun_out/final/bench_c3_pair_f16mlp.json 2> gpurun_out/final/bench_c3_pair_f16mlp.err
  for f in c3_pair c3_pair_f16mlp; do tail -n 1 gpurun_out/final/bench_$f.json | cut -c1-200; done
  mkdir -p gpurun_out/cells2
  for sl in 2 6 7; do for pr in f32 f16mlp; do
    python3 tools/hip_cell_sampler.py --cell it200 --slice $sl --runs 64 --precision $pr --out gpurun_out/cells2/it200_s${sl}_$pr.npz | tee -a gpurun_out/cells2/summary.txt
  done; done
else
  python3 bench.py --workload c3 --batch 64 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/final/bench_c3_serial.json 2> gpurun_out/final/bench_c3_serial.err
  python3 bench.py --workload c3 --batch 64 --steps 1 --warmup 0 --no-cpu-baseline --pair > gpurun_out/final/bench_c3_pair.json 2> gpurun_out/final/bench_c3_pair.err
  python3 bench.py --workload c3 --batch 64 --steps 1 --warmup 0 --no-cpu-baseline --pair --precision f16mlp > gpurun_out/final/bench_c3_pair_f16mlp.json 2> gpurun_out/final/bench_c3_pair_f16mlp.err
  python3 bench.py --workload c3 --batch 64 --steps 1 --warmup 0 --no-cpu-baseline --precision f16mlp > gpurun_out/final/bench_c3_serial_f16mlp.json 2> gpurun_out/final/bench_c3_serial_f16mlp.err
  for f in c3_serial c3_pair c3_pair_f16mlp c3_serial_f16mlp; do tail -n 1 gpurun_out/final/bench_$f.json | cut -c1-200; done
  bash tools/timeline_pair.sh > /dev/null || true
fi
