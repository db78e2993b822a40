set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/tl
for mode in f32 f16mlp bf16x2; do
  rm -rf gpurun_out/tl/$mode
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl/$mode -- python3 bench.py --iters 40 --steps 1 --warmup 0 --no-cpu-baseline --no-alt-precision --precision $mode > gpurun_out/tl/$mode.log 2>&1
  python3 tools/timeline.py gpurun_out/tl/$mode/*/*_kernel_trace.csv 20 > gpurun_out/tl/$mode.txt
  rm -rf gpurun_out/tl/$mode
done
bash tools/timeline_pair.sh > /dev/null
tail -n 3 gpurun_out/tl/f32.txt gpurun_out/tl/f16mlp.txt
