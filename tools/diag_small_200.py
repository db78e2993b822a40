"""HIP side of the small-scale 200-iteration experiment (96x96, 3 groups, slice 11): N runs, median PSNR / loss of the
last 21 iterations (GPU box).     python tools/diag_small_200.py [N=32] [--mlp-fp16 | --bf16x2] [--seeds=K]
--seeds=K: run r starts from init_params seed 2001 + r % K instead of 1337."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import miccai24_immoco_amd as pkg
from miccai24_immoco_amd.models.immoco import get_solver
from miccai24_immoco_amd.utils.motion_utils import masks_to_col_group
from oracle import synth_cpu
from _stats import hip_psnr_samples, summarize
a = [x for x in sys.argv[1:] if not x.startswith("--")]
N = int(a[0]) if a else 32
s_ = synth_cpu.make_slice(96, 96, 3, 11)
k, lines = s_["kspace"].cuda(), s_["lines"].cuda()
masks = pkg.extract_movement_groups(lines, make_list=True)
gt = s_["gt"].abs()
sol = get_solver(torch.device("cuda", 0), 96, 96, int(masks.shape[0]), mlp_fp16=("bf16x2" if "--bf16x2" in sys.argv else "--mlp-fp16" in sys.argv))
kin, cg = k / k.abs().max() * 16000, masks_to_col_group(masks)
K = next((int(x.split("=")[1]) for x in sys.argv if x.startswith("--seeds=")), 0)
ps_, ls_ = [], []
for r in range(N):
    ps, loss = hip_psnr_samples(sol, kin, cg, gt, 200, list(range(179, 200)), seed=(2001 + r % K if K else 1337))
    ps_.append(float(np.median(list(ps.values())))); ls_.append(float(np.median(loss[179:])))
print("HIP 96x96x3 slice 11, 200 iterations,", N, "runs", [x for x in sys.argv if x.startswith("--")],
      ": median-of-last-21 PSNR mean %.3f sd %.3f se %.3f | loss mean %.4f" % (*summarize(ps_), float(np.mean(ls_))))
if K:
    print("per-seed means:", [round(float(np.mean(ps_[j::K])), 3) for j in range(K)])
else:
    print(np.round(ps_, 2).tolist())
