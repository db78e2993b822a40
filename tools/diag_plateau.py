"""Plateau PSNR (per run: median over iterations 600, 625, ..., 1375 of the 3000-iteration schedule) and blow-up events of
N HIP solves of C2 slice 1 (GPU box).      python tools/diag_plateau.py [N=24] [--mlp-fp16]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import miccai24_immoco_amd as pkg
from miccai24_immoco_amd import synth
from miccai24_immoco_amd.models.immoco import get_solver
from miccai24_immoco_amd.utils.motion_utils import masks_to_col_group
from oracle import synth_cpu
from _stats import hip_psnr_samples, summarize
a = [x for x in sys.argv[1:] if not x.startswith("--")]
N = int(a[0]) if a else 24
s_ = synth_cpu.make_slice(320, 320, 10, 1)
k, lines = s_["kspace"].cuda(), s_["lines"].cuda()
masks = pkg.extract_movement_groups(lines, make_list=True)
gt = synth.phantom(320, 320, 1001).abs()
sol = get_solver(torch.device("cuda", 0), 320, 320, 10, mlp_fp16=("bf16x2" if "--bf16x2" in sys.argv else "--mlp-fp16" in sys.argv))
kin, cg = k / k.abs().max() * 16000, masks_to_col_group(masks)
grid = list(range(600, 1400, 25))
h = []
for r in range(N):
    ps, loss = hip_psnr_samples(sol, kin, cg, gt, 3000, grid)
    h.append(float(np.median([ps[t] for t in grid])))
print("plateau PSNR", [x for x in sys.argv if x.startswith("--")], "mean %.3f sd %.3f se %.3f" % summarize(h), np.round(sorted(h), 2).tolist())
