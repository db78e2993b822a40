"""Every-iteration PSNR / loss of HIP solves of C2 slice 1 over iterations 250..900: where a run leaves the 39.6 dB plateau,
how fast, and what the loss does there (GPU box).   python tools/diag_drop.py [runs=12]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import miccai24_immoco_amd as pkg
from miccai24_immoco_amd import synth
from miccai24_immoco_amd.models.immoco import get_solver, lambda_schedule
from miccai24_immoco_amd.utils.evaluate import crop_psnr
from miccai24_immoco_amd.utils.motion_utils import masks_to_col_group
from oracle import synth_cpu
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 12
s_ = synth_cpu.make_slice(320, 320, 10, 1)
k, lines = s_["kspace"].cuda(), s_["lines"].cuda()
masks = pkg.extract_movement_groups(lines, make_list=True)
gt = synth.phantom(320, 320, 1001).abs()
sol = get_solver(torch.device("cuda", 0), 320, 320, 10)
kin = k / k.abs().max() * 16000
cg = masks_to_col_group(masks)
lam = lambda_schedule(3000, 1e-2)
t0, t1 = 250, 900
for r in range(runs):
    pi, pm = sol.init_params()
    ai, am = torch.zeros(2 * pi.numel(), device="cuda"), torch.zeros(2 * pm.numel(), device="cuda")
    sol.solve(kin, cg, pi, pm, ai, am, t0, 1e-2, lam[:t0])
    ps, ls = [], []
    for t in range(t0, t1):
        img, _, l = sol.solve(kin, cg, pi, pm, ai, am, 1, 1e-2, lam[t:t + 1], step0=t, want_loss=True)
        ps.append(crop_psnr(img.abs().cpu(), gt)); ls.append(float(l[0]))
    ps, ls = np.array(ps), np.array(ls)
    sm = np.convolve(ps, np.ones(10) / 10, mode="valid")          # 10-iteration mean (removes the period-2 ripple)
    d = sm[20:] - sm[:-20]                                         # change over 20 iterations
    j = int(d.argmin())
    print(f"run {r}: smoothed PSNR start {sm[0]:.2f} end {sm[-1]:.2f} min {sm.min():.2f}; steepest 20-iteration decline {d[j]:.2f} dB at {t0 + j}..{t0 + j + 20}")
    if d[j] < -1.5:
        a, b = max(0, j - 4), min(len(ps), j + 34)
        print("   psnr", np.round(ps[a:b], 1).tolist())
        print("   loss", np.round(ls[a:b], 2).tolist())
