"""Cost of re-capturing the iteration graphs when a solve is handed other buffers: short solves
(16 iterations) with the SAME parameter / Adam buffers (graph replayed) vs freshly allocated ones (single-iteration
graph + 8-iteration graph captured and instantiated again).  GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import miccai24_immoco_amd as pkg
from miccai24_immoco_amd import synth
from miccai24_immoco_amd.models.immoco import get_solver, lambda_schedule
from miccai24_immoco_amd.utils.motion_utils import masks_to_col_group
dev = torch.device("cuda", 0)
s = synth.make_slice(320, 320, 10, 1, device=dev)
masks = pkg.extract_movement_groups(s["lines"], make_list=True)
sol = get_solver(dev, 320, 320, int(masks.shape[0]))
k = s["kspace"]; kin = k / k.abs().max() * 16000
cg = masks_to_col_group(masks)
lam = lambda_schedule(3000, 1e-2)[:16]
def bufs():
    pi, pm = sol.init_params()
    return pi, pm, torch.zeros(2 * pi.numel(), device=dev), torch.zeros(2 * pm.numel(), device=dev)
def run(b):
    torch.cuda.synchronize(); t = time.perf_counter()
    sol.solve(kin, cg, *b, 16, 1e-2, lam)
    torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3
b = bufs(); run(b)
same = sorted(run(b) for _ in range(10))[5]
keep = []
fresh = []
for _ in range(10):
    nb = bufs(); keep.append(nb); fresh.append(run(nb))
fresh = sorted(fresh)[5]
print(f"16-iteration solve: same buffers {same:.2f} ms, new buffers (re-capture of both graphs) {fresh:.2f} ms -> {fresh - same:.2f} ms per re-capture "
      f"= {100 * (fresh - same) / 3757:.3f} % of a 3000-iteration solve")
