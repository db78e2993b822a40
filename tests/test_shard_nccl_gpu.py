"""RCCL path of the slice sharding on the device: `solve_sharded` + `gather_images` through the
`nccl` backend (= RCCL on ROCm) at world size 1 on cuda:0 - what `bench.py` does for N > 1 (and for N = 1 under
BENCH_FORCE_DIST).  Runs in a child process so that the process group never lives in the pytest process; a real
HIP solve feeds the gather.  No multi-GPU node is available to the build: the N > 1 data path is the same code with
world > 1 (covered on CPU by tests/test_shard_gloo.py)."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent("""
    import os, sys, socket
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", device_id=dev)
    import miccai24_immoco_amd as pkg
    from miccai24_immoco_amd import synth
    from miccai24_immoco_amd.shard import solve_sharded, gather_images
    sl = [synth.make_slice(64, 64, 3, i, device=dev) for i in range(3)]
    masks = [pkg.extract_movement_groups(s["lines"], make_list=True) for s in sl]
    solved = {}
    def solve(i):
        solved[i] = pkg.imcoco_motion_correction(sl[i]["kspace"], masks[i], iters=20)[0].detach()
        return solved[i]
    out = solve_sharded(3, solve)
    torch.cuda.synchronize()
    ref = torch.stack([solved[i] for i in range(3)])
    assert out.shape == ref.shape and out.dtype == torch.complex64 and out.is_cuda
    assert torch.equal(out, ref), "all_gather_into_tensor round trip changed the images"
    one = gather_images(ref, 3, dst=0)          # the gather-to-rank-0 flavour
    assert torch.equal(one, ref)
    assert dist.get_backend() == "nccl"
    dist.barrier(); dist.destroy_process_group()
    print("NCCL_WORLD1_OK", float(ref.abs().mean()))
    """) % ROOT


@pytest.mark.gpu
def test_gather_images_over_rccl_world1_on_device():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "NCCL_WORLD1_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
