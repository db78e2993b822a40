# round-4 closing run: the full -m gpu suite, then the default bench line and the kernel-trace evidence of the same code.
# A step that was killed or timed out ends the run; a failed assertion does not (the evidence is still wanted).
set -o pipefail
mkdir -p gpurun_out/suite gpurun_out/final2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 780 python3 -m pytest tests -m gpu -q -rxX --durations=12 > gpurun_out/suite/r4_final.log 2>&1
rc=$?
tail -n 25 gpurun_out/suite/r4_final.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 400 python3 bench.py > gpurun_out/final2/bench_default.json 2> gpurun_out/final2/bench_default.err || exit 1
tail -n 1 gpurun_out/final2/bench_default.json | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final2/kt_default -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-precision > gpurun_out/final2/kt_default.log 2>&1 || exit 1
rm -f gpurun_out/final2/kt_default/*/*_kernel_trace.csv
python3 bench.py --precision f16mlp --steps 1 --warmup 1 --no-cpu-baseline --no-alt-precision > gpurun_out/final2/bench_f16mlp.json 2> gpurun_out/final2/bench_f16mlp.err || exit 1
mkdir -p gpurun_out/tl
for mode in f32 f16mlp; do
  rm -rf gpurun_out/tl/$mode
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl/$mode -- python3 bench.py --iters 40 --steps 1 --warmup 0 --no-cpu-baseline --no-alt-precision --precision $mode > gpurun_out/tl/$mode.log 2>&1 || exit 1
  python3 tools/timeline.py gpurun_out/tl/$mode/*/*_kernel_trace.csv 20 > gpurun_out/final2/timeline_$mode.txt
  rm -rf gpurun_out/tl/$mode
done
exit $rc
