"""End-of-solve statistics of config C2, slice 1 (tests/golden/c2_slice1_input.npz): N HIP solves of 3000 iterations,
PSNR of the last forward at iterations 1400 (lambda_GE still > 0) and 2999, next to the CPU oracle's full records.
    python tools/diag_c2_end_psnr.py [N=20] [slice_idx]
With a slice index other than 1 the input is regenerated with the CPU generator of the tests (oracle/synth_cpu.py, what
tools/oracle_c2.py runs on) and only the HIP statistics are printed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import miccai24_immoco_amd as pkg
from miccai24_immoco_amd import synth
from miccai24_immoco_amd.models.immoco import get_solver, lambda_schedule
from miccai24_immoco_amd.utils.evaluate import crop_psnr
from miccai24_immoco_amd.utils.motion_utils import masks_to_col_group
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sl = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rec = np.load(os.path.join(ROOT, "tests", "golden", "c2_oracle_slice1_3000it.npz")) if sl == 1 else None
if sl == 1:
    g = np.load(os.path.join(ROOT, "tests", "golden", "c2_slice1_input.npz"))
    k = torch.from_numpy(g["kspace"]).cuda()
    lines = torch.from_numpy(g["lines"]).cuda()
else:
    from oracle import synth_cpu
    s_ = synth_cpu.make_slice(320, 320, 10, sl)
    k, lines = s_["kspace"].cuda(), s_["lines"].cuda()
masks = pkg.extract_movement_groups(lines, make_list=True)
gt = synth.phantom(320, 320, 1000 + sl).abs()
sol = get_solver(torch.device("cuda", 0), 320, 320, int(masks.shape[0]))
kin = k / k.abs().max() * 16000
cg = masks_to_col_group(masks)
lam = lambda_schedule(3000, 1e-2)
rows = []
for r in range(n):
    pi, pm = sol.init_params()
    ai, am = torch.zeros(2 * pi.numel(), device="cuda"), torch.zeros(2 * pm.numel(), device="cuda")
    img1, _, l1 = sol.solve(kin, cg, pi, pm, ai, am, 1401, 1e-2, lam[:1401], want_loss=True)
    img2, _, l2 = sol.solve(kin, cg, pi, pm, ai, am, 1599, 1e-2, lam[1401:], step0=1401, want_loss=True)
    rows.append((crop_psnr(img1.abs().cpu(), gt), crop_psnr(img2.abs().cpu(), gt), float(l1[-1]), float(l2[-1])))
    print(r, "psnr@1400 %.2f psnr@2999 %.2f loss@1400 %.3f loss@2999 %.5f" % rows[-1], flush=True)
a = np.array(rows)
print("HIP   psnr@1400 mean %.2f sd %.2f | psnr@2999 mean %.2f sd %.2f min %.2f max %.2f" %
      (a[:, 0].mean(), a[:, 0].std(), a[:, 1].mean(), a[:, 1].std(), a[:, 1].min(), a[:, 1].max()))
if rec is not None:
    it = rec["oracle_psnr_iters"]
    print("oracle psnr@1400", rec["oracle_psnr"][:, list(it).index(1400)].round(2), "psnr@2999", rec["oracle_psnr"][:, -1].round(2))
