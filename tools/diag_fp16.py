"""Diagnostic: PSNR of the fp32 and fp16-table solver modes over several C2 slices (GPU)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import miccai24_immoco_amd as pkg
from miccai24_immoco_amd import synth
from miccai24_immoco_amd.utils.evaluate import crop_psnr
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
res = {False: [], True: []}
for i in range(8):
    sl = synth.make_slice(320, 320, 10, i, device="cuda")
    sl["masks"] = pkg.extract_movement_groups(sl["lines"], make_list=True)
    for f16 in (False, True):
        img, _ = pkg.imcoco_motion_correction(sl["kspace"], sl["masks"], iters=iters, table_fp16=f16)
        res[f16].append(round(float(crop_psnr(img.abs(), sl["gt"].abs())), 2))
for f16 in (False, True):
    print("fp16" if f16 else "fp32", "iters", iters, res[f16], "median", np.median(res[f16]), flush=True)
