"""Diagnostic: run-to-run PSNR spread of the HIP solver on the golden cases (GPU)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import miccai24_immoco_amd as pkg
from oracle import immoco_oracle as orc
from conftest import expand_masks
g = dict(np.load(os.path.join(ROOT, "tests/golden/solver.npz")))
for tag in ("c32", "c48"):
    H = g[f"{tag}_gt"].shape[0]
    masks = expand_masks(g[f"{tag}_masks_row0"], H).cuda()
    gt = torch.from_numpy(g[f"{tag}_gt"]).abs()
    ksp = torch.from_numpy(g[f"{tag}_ksp"]).cuda()
    ref = orc.crop_psnr(torch.from_numpy(np.abs(g[f"{tag}_image_prior"])), gt)
    for iters in (int(g[f"{tag}_iters"]), 100, 300):
        for atomic in (False, True):
            ps = []
            for r in range(5):
                img, _ = pkg.imcoco_motion_correction(ksp, masks, iters=iters, atomic_scatter=atomic)
                ps.append(orc.crop_psnr(img.abs().cpu(), gt))
            print(tag, "iters", iters, "atomic" if atomic else "csr", "ref(golden iters)", round(ref, 3), [round(p, 3) for p in ps], flush=True)
