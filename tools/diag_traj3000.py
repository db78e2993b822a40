"""Diagnostic: PSNR / loss trajectory of a 3000-iteration HIP solve on C2 slice 1, in segments."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import miccai24_immoco_amd as pkg
from miccai24_immoco_amd import synth
from miccai24_immoco_amd.models.immoco import get_solver, lambda_schedule
from miccai24_immoco_amd.utils.motion_utils import masks_to_col_group
from miccai24_immoco_amd.utils.evaluate import crop_psnr
idx = int(sys.argv[1]) if len(sys.argv) > 1 else 1
s = synth.make_slice(320, 320, 10, idx)
masks = pkg.extract_movement_groups(s["lines"].cuda(), make_list=True)
sol = get_solver("cuda", 320, 320, masks.shape[0])
k = s["kspace"].cuda(); kin = k / k.abs().max() * 16000
cg = masks_to_col_group(masks)
pi, pm = sol.init_params()
ai = torch.zeros(2 * pi.numel(), device="cuda"); am = torch.zeros(2 * pm.numel(), device="cuda")
lam = lambda_schedule(3000, 1e-2)
seg = 125
for a in range(0, 3000, seg):
    img, kf, loss = sol.solve(kin, cg, pi, pm, ai, am, seg, 1e-2, lam[a:a + seg], step0=a, want_loss=True)
    lh = loss.cpu().numpy()
    print(a + seg - 1, "loss %.4g" % lh[-1], "lambda %.3g" % lam[a + seg - 1], "psnr %.3f" % crop_psnr(img.abs().cpu(), s["gt"].abs()), flush=True)
