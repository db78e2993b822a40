"""Diagnostic: HIP solver on C2 slice 1, 300 iterations, vs the CPU oracle's trajectory
(tests/golden/c2_oracle_slice1_300it.npz, produced by tools/oracle_c2.py)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import miccai24_immoco_amd as pkg
from oracle import synth_cpu as synth
from miccai24_immoco_amd.utils.evaluate import crop_psnr
g = np.load(os.path.join(ROOT, "tests/golden/c2_oracle_slice1_300it.npz"))
ol = g["loss"].astype(np.float64)
s = synth.make_slice(320, 320, 10, 1)
masks = pkg.extract_movement_groups(s["lines"].cuda(), make_list=True)
its = [0, 1, 2, 5, 10, 20, 50, 100, 150, 200, 250, 299]
print("oracle loss", ["%.4g" % ol[i] for i in its], "psnr@299 %.3f" % g["psnr"][-1])
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    img, _, loss = pkg.imcoco_motion_correction(s["kspace"].cuda(), masks, iters=300, return_loss=True)
    lh = loss.cpu().numpy().astype(np.float64)
    print("hip    loss", ["%.4g" % lh[i] for i in its], "psnr %.3f" % crop_psnr(img.abs().cpu(), s["gt"].abs()), flush=True)
