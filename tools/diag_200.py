"""Median-of-last-21 PSNR of N HIP solves of the 200-iteration schedule on one C2 slice (GPU box).
    python tools/diag_200.py <slice> [N=64] [--mlp-fp16 | --bf16x2] [--seeds=K]
--seeds=K: run r starts from init_params seed 2001 + r % K instead of the reference's fixed 1337 (per-seed means printed)."""
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
sl, N = int(a[0]), (int(a[1]) if len(a) > 1 else 64)
s_ = synth_cpu.make_slice(320, 320, 10, sl)
k, lines = s_["kspace"].cuda(), s_["lines"].cuda()
masks = pkg.extract_movement_groups(lines, make_list=True)
gt = synth.phantom(320, 320, 1000 + sl).abs()
sol = get_solver(torch.device("cuda", 0), 320, 320, int(masks.shape[0]),
                 mlp_fp16=("bf16x2" if "--bf16x2" in sys.argv else "--mlp-fp16" in sys.argv))
kin, cg = k / k.abs().max() * 16000, masks_to_col_group(masks)
K = next((int(x.split("=")[1]) for x in sys.argv if x.startswith("--seeds=")), 0)
h, by_seed = [], {}
for r in range(N):
    seed = 2001 + r % K if K else 1337
    ps, loss = hip_psnr_samples(sol, kin, cg, gt, 200, list(range(179, 200)), seed=seed)
    h.append(float(np.median(list(ps.values()))))
    by_seed.setdefault(seed, []).append(h[-1])
if K:
    for sd in sorted(by_seed):
        print("init seed %d: mean %.3f sd %.3f se %.3f (%d runs)" % (sd, *summarize(by_seed[sd]), len(by_seed[sd])))
    print("mean over seeds of the per-seed means: %.3f" % np.mean([np.mean(v) for v in by_seed.values()]))
print("slice", sl, [x for x in sys.argv if x.startswith("--")], N, "runs: median-of-last-21 PSNR mean %.3f sd %.3f se %.3f" % summarize(h),
      "median %.2f" % np.median(h), "deciles", np.round(np.quantile(h, np.linspace(0.1, 0.9, 9)), 2).tolist())
