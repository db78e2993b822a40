"""When do HIP runs of C2 slice 1 leave the ~39.6 dB plateau?  N runs sampled every 25 iterations from 100 to 1400; for the
runs whose 600..1375 median is below 38 dB: the PSNR trace and the loss events (GPU box).
    python tools/diag_lowbasin.py [N=40] [--mlp-fp16 | --bf16x2] [--atomic]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import miccai24_immoco_amd as pkg
from miccai24_immoco_amd import synth
from miccai24_immoco_amd.models.immoco import get_solver
from miccai24_immoco_amd.utils.motion_utils import masks_to_col_group
from oracle import synth_cpu
from _stats import hip_psnr_samples
a = [x for x in sys.argv[1:] if not x.startswith("--")]
N = int(a[0]) if a else 40
s_ = synth_cpu.make_slice(320, 320, 10, 1)
k, lines = s_["kspace"].cuda(), s_["lines"].cuda()
masks = pkg.extract_movement_groups(lines, make_list=True)
gt = synth.phantom(320, 320, 1001).abs()
sol = get_solver(torch.device("cuda", 0), 320, 320, 10, atomic_scatter="--atomic" in sys.argv,
                 mlp_fp16=("bf16x2" if "--bf16x2" in sys.argv else "--mlp-fp16" in sys.argv))
kin, cg = k / k.abs().max() * 16000, masks_to_col_group(masks)
grid = list(range(100, 1400, 25))


def events(l, a, b, thr=1.5):
    ev, t = [], a
    while t < b:
        med = np.median(l[t - 20:t])
        if l[t] > thr * med:
            ev.append((t, round(float(l[t] / med), 1))); t += 40
        else:
            t += 1
    return ev


low = 0
for r in range(N):
    ps, loss = hip_psnr_samples(sol, kin, cg, gt, 3000, grid)
    tr = np.array([ps[t] for t in grid])
    plat = float(np.median(tr[[grid.index(t) for t in range(600, 1400, 25)]]))
    if plat < 38.0:
        low += 1
        first = next((t for t, v in zip(grid, tr) if t >= 300 and v < 37.5), None)
        print(f"run {r}: plateau {plat:.2f}; first sample < 37.5 dB at {first}; loss events {events(loss.astype(float), 100, 1375)}")
        print("   trace 100..1375:", np.round(tr, 1).tolist())
print("flags", [x for x in sys.argv if x.startswith("--")], "low-plateau runs:", low, "of", N)
