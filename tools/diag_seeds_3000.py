"""The metric's 3000-iteration solve of C2 slice 1 from K initialisations (init_params seeds 2001 ... 2000+K), R runs
each, in one arithmetic: per run the plateau level (median PSNR over iterations 600, 625, ..., 1375; lambda_GE > 0) and
the end-of-solve level (median over 2900, 2925, ..., 2999; lambda_GE = 0 since iteration 1500).  Per-seed means and the
mean over seeds; traces dumped to gpurun_out/seeds3000_<tag>.npz (GPU box).
    python tools/diag_seeds_3000.py [K=8] [R=6] [--mlp-fp16 | --bf16x2] [--tag=NAME]"""
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
K, R = (int(a[0]) if a else 8), (int(a[1]) if len(a) > 1 else 6)
tag = next((x.split("=")[1] for x in sys.argv if x.startswith("--tag=")), "f32")
s_ = synth_cpu.make_slice(320, 320, 10, 1)
k, lines = s_["kspace"].cuda(), s_["lines"].cuda()
masks = pkg.extract_movement_groups(lines, make_list=True)
gt = synth.phantom(320, 320, 1001).abs()
sol = get_solver(torch.device("cuda", 0), 320, 320, 10,
                 mlp_fp16=("bf16x2" if "--bf16x2" in sys.argv else "--mlp-fp16" in sys.argv))
kin, cg = k / k.abs().max() * 16000, masks_to_col_group(masks)
g1, g2 = list(range(600, 1400, 25)), [2900, 2925, 2950, 2975, 2999]
seeds, plat, end = [], [], []
for r in range(R):
    for j in range(K):
        ps, _ = hip_psnr_samples(sol, kin, cg, gt, 3000, g1 + g2, seed=2001 + j)
        seeds.append(2001 + j)
        plat.append(float(np.median([ps[t] for t in g1])))
        end.append(float(np.median([ps[t] for t in g2])))
seeds, plat, end = np.array(seeds), np.array(plat), np.array(end)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"seeds3000_{tag}.npz"), seed=seeds, plateau=plat, end=end)
for sd in sorted(set(seeds.tolist())):
    m = seeds == sd
    print("init seed %d: plateau mean %.2f (low %d of %d) | end-of-solve mean %.2f sd %.2f" %
          (sd, plat[m].mean(), int((plat[m] < 38).sum()), int(m.sum()), end[m].mean(), end[m].std(ddof=1)))
print(tag, "mean over seeds of per-seed means: plateau %.3f, end-of-solve %.3f; all runs: plateau %.3f +- %.3f (low %d of %d), end %.3f +- %.3f"
      % (np.mean([plat[seeds == sd].mean() for sd in set(seeds.tolist())]), np.mean([end[seeds == sd].mean() for sd in set(seeds.tolist())]),
         *summarize(plat)[::2], int((plat < 38).sum()), len(plat), *summarize(end)[::2]))
