"""Per-kernel times of one solver iteration (immoco_solver_profile: HIP events on the solver stream, eager,
serial) plus the graph-replayed iteration time.  GPU box.
    python tools/phase_times.py [H W nM] [--fp16] [--parts N] [--iters 300]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import miccai24_immoco_amd as pkg
from miccai24_immoco_amd import synth
from miccai24_immoco_amd.models.immoco import get_solver, lambda_schedule
from miccai24_immoco_amd.utils.motion_utils import masks_to_col_group
ap = argparse.ArgumentParser()
ap.add_argument("shape", nargs="*", type=int, default=[320, 320, 10])
ap.add_argument("--fp16", action="store_true")
ap.add_argument("--parts", type=int, default=0)
ap.add_argument("--iters", type=int, default=300)
a = ap.parse_args()
H, W, nM = a.shape
dev = torch.device("cuda", 0)
s = synth.make_slice(H, W, nM, 1, device=dev)
masks = pkg.extract_movement_groups(s["lines"], make_list=True)
nM = int(masks.shape[0])
sol = get_solver(dev, H, W, nM, True, False, a.parts, 0, a.fp16)
k = s["kspace"]
kin = k / k.abs().max() * 16000
cg = masks_to_col_group(masks)
pi, pm = sol.init_params()
ai, am = torch.zeros(2 * pi.numel(), device=dev), torch.zeros(2 * pm.numel(), device=dev)
sol.profile(kin, cg, pi, pm, ai, am, reps=3)
ph = sol.profile(kin, cg, pi, pm, ai, am, reps=20)
for n, ms in ph:
    print(f"{n:22s} {ms:.4f}")
print(f"{'sum':22s} {sum(ms for _, ms in ph):.4f}")
lam = lambda_schedule(3000, 1e-2)[:a.iters]
sol.solve(kin, cg, pi, pm, ai, am, a.iters, 1e-2, lam)
torch.cuda.synchronize()
t0 = time.perf_counter()
sol.solve(kin, cg, pi, pm, ai, am, a.iters, 1e-2, lam)
torch.cuda.synchronize()
print(f"graph iteration ms     {(time.perf_counter() - t0) / a.iters * 1e3:.4f}  (entries motion plan: "
      f"{int(pkg._lib.lib().immoco_solver_plan_entries(sol.handle, 1))})")
