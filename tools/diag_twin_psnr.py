"""Diagnostic: PSNR spread over repeated 300-iteration solves of C2 slices (run once per IMMOCO_CSR_NO_TWIN setting)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import miccai24_immoco_amd as pkg
from miccai24_immoco_amd import synth
from miccai24_immoco_amd.utils.evaluate import crop_psnr
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
allp = []
for idx in (1, 2, 3):
    sl = synth.make_slice(320, 320, 10, idx, device="cuda")
    masks = pkg.extract_movement_groups(sl["lines"], make_list=True)
    ps, fl = [], []
    for rep in range(6):
        img, _, loss = pkg.imcoco_motion_correction(sl["kspace"], masks, iters=iters, return_loss=True)
        ps.append(round(float(crop_psnr(img.abs(), sl["gt"].abs())), 2))
        fl.append(round(float(loss[-1]), 4))
    allp += ps
    print("twin" if not os.environ.get("IMMOCO_CSR_NO_TWIN") else "no-twin", "slice", idx, "psnr", ps, "final loss", fl, flush=True)
print("median", np.median(allp), "mean", np.mean(allp))
