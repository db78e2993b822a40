"""PSNR statistics of N HIP solves of one C2-shaped slice next to the CPU oracle's draws (GPU box).

    python tools/diag_stats.py <slice_idx> <sched_iters: 200 | 3000> <n_runs> [--mlp-fp16] [--table-fp16]

sched 200 (the reference script's own setting, /root/reference/src/test/test_immoco.py:65-72): every run is sampled at
iterations 179 ... 199; statistics of the FINAL forward (what the reference returns) and of the median over the last 21.
sched 3000 (the metric's): sampled at 1350, 1375, 1400, 1425, 1450 (lambda_GE > 0; the oracle records hold PSNR every
25 iterations) and 2950 ... 2999 every 25 + 2999.  Oracle side: tests/golden/c2_oracle_200it_draws.npz /
c2_oracle_slice1_3000it.npz (+ c2_oracle_slice1_redraw1400.npz)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import miccai24_immoco_amd as pkg
from miccai24_immoco_amd import synth
from miccai24_immoco_amd.models.immoco import get_solver
from miccai24_immoco_amd.utils.motion_utils import masks_to_col_group
from _stats import hip_psnr_samples, summarize, delta_with_se
sl, sched, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mlp16, tab16 = "--mlp-fp16" in sys.argv, "--table-fp16" in sys.argv
G = os.path.join(ROOT, "tests", "golden")
from oracle import synth_cpu          # the oracle records' input generator (test infrastructure)
s_ = synth_cpu.make_slice(320, 320, 10, sl)
k, lines = s_["kspace"].cuda(), s_["lines"].cuda()
masks = pkg.extract_movement_groups(lines, make_list=True)
gt = synth.phantom(320, 320, 1000 + sl).abs()
sol = get_solver(torch.device("cuda", 0), 320, 320, int(masks.shape[0]), table_fp16=tab16, mlp_fp16=mlp16)
kin = k / k.abs().max() * 16000
cg = masks_to_col_group(masks)
samples = list(range(179, 200)) if sched == 200 else [1350, 1375, 1400, 1425, 1450, 2900, 2925, 2950, 2975, 2999]
rows = []
for r in range(n):
    ps, loss = hip_psnr_samples(sol, kin, cg, gt, sched, samples)
    rows.append([ps[t] for t in samples])
    print(r, " ".join(f"{ps[t]:.2f}" for t in samples), flush=True)
a = np.array(rows)
tag = f"slice {sl} sched {sched} mlp_fp16={mlp16} table_fp16={tab16} runs={n}"
if sched == 200:
    fin, med = a[:, -1], np.median(a, axis=1)
    print(tag, "| final: mean %.3f sd %.3f se %.3f | median of last 21: mean %.3f sd %.3f se %.3f" % (*summarize(fin), *summarize(med)))
    f = os.path.join(G, "c2_oracle_200it_draws.npz")
    if os.path.exists(f):
        g = np.load(f)
        key = f"s{sl}_psnr"
        if key in g:
            o = g[key]
            print("oracle final", np.round(o[:, -1], 2), "median21", np.round(np.median(o[:, 179:200], axis=1), 2))
            print("delta final %.3f +- %.3f (var ratio %.2f) | delta median21 %.3f +- %.3f (var ratio %.2f)" %
                  (*delta_with_se(fin, o[:, -1]), *delta_with_se(med, np.median(o[:, 179:200], axis=1))))
else:
    mid, end = np.median(a[:, :5], axis=1), np.median(a[:, 5:], axis=1)
    print(tag, "| @1400 single: mean %.3f sd %.3f | median(1350..1450): mean %.3f sd %.3f se %.3f | end median: mean %.3f sd %.3f | end single %.3f sd %.3f"
          % (*summarize(a[:, 2])[:2], *summarize(mid), *summarize(end)[:2], *summarize(a[:, -1])[:2]))
    if sl == 1:
        rec = np.load(os.path.join(G, "c2_oracle_slice1_3000it.npz"))
        it = list(rec["oracle_psnr_iters"])
        o = rec["oracle_psnr"]
        om = np.median(o[:, [it.index(t) for t in samples[:5]]], axis=1)
        oe = np.median(o[:, [it.index(t) for t in samples[5:]]], axis=1)
        print("oracle @1400 single", np.round(o[:, it.index(1400)], 2), "median", np.round(om, 2), "end median", np.round(oe, 2))
        print("delta @1400 single %.3f +- %.3f (var ratio %.2f) | median %.3f +- %.3f (var ratio %.2f) | end median %.3f +- %.3f (var ratio %.2f)"
              % (*delta_with_se(a[:, 2], o[:, it.index(1400)]), *delta_with_se(mid, om), *delta_with_se(end, oe)))
