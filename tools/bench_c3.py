"""Config 3 scan (GPU box): B slices of the C2 shape through immoco_solver_solve_batch with `lanes` slices in
flight; prints ms per slice-iteration for each lane count.  Set GPU_MAX_HW_QUEUES=12 in the environment, otherwise
HIP folds the lanes' streams onto 4 hardware queues and they run one after the other anyway.
    GPU_MAX_HW_QUEUES=12 python tools/bench_c3.py [--B 8] [--iters 300] [--lanes 1,2,3]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import miccai24_immoco_amd as pkg
from miccai24_immoco_amd import synth
from miccai24_immoco_amd.models.immoco import _SOLVERS
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=8)
ap.add_argument("--iters", type=int, default=300)
ap.add_argument("--lanes", default="1,2,3")
a = ap.parse_args()
dev = torch.device("cuda", 0)
sl = [synth.make_slice(320, 320, 10, i, device=dev) for i in range(a.B)]
masks = [pkg.extract_movement_groups(s["lines"], make_list=True) for s in sl]
ksp = torch.stack([s["kspace"] for s in sl])
lam = 1e-2
for lanes in [int(x) for x in a.lanes.split(",")]:
    for waves in (0,):
        def run():
            return pkg.imcoco_motion_correction_batch(ksp, masks, iters=a.iters, lanes=lanes)
        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        imgs, _ = run()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"lanes {lanes}: {dt / (a.B * a.iters) * 1e3:.4f} ms per slice-iteration "
              f"({a.B} slices x {a.iters} it in {dt:.2f} s; finite {bool(torch.isfinite(torch.view_as_real(imgs)).all())})",
              flush=True)
        for k in list(_SOLVERS):
            _SOLVERS.pop(k).close()
        torch.cuda.empty_cache()
