"""Diagnostic: per-iteration loss of the HIP solver vs the CPU oracle on golden case c48 (GPU)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import miccai24_immoco_amd as pkg
from oracle import immoco_oracle as orc
from conftest import expand_masks
g = dict(np.load(os.path.join(ROOT, "tests/golden/solver.npz")))
tag = "c48"; H = 48; iters = 30
masks = expand_masks(g[f"{tag}_masks_row0"], H)
gt = torch.from_numpy(g[f"{tag}_gt"]).abs()
ksp = torch.from_numpy(g[f"{tag}_ksp"])
hist = []
ref = orc.OracleIMMoCo(masks, image_inr=orc.OracleINR(2, 2, orc.encoding_config, orc.network_config),
                       motion_inr=orc.OracleINR(3, 2, orc.encoding_config, orc.mot_network_config))
img_ref, _ = orc.oracle_motion_correction(ksp, masks, iters=iters, model=ref, loss_hist=hist)
print("impl", os.environ.get("IMMOCO_MLP_IMPL", "mfma"), "oracle psnr", orc.crop_psnr(img_ref.detach().abs(), gt))
hist = np.array(hist)
for r in range(4):
    img, _, loss = pkg.imcoco_motion_correction(ksp.cuda(), masks.cuda(), iters=iters, return_loss=True)
    lh = loss.cpu().numpy().astype(np.float64)
    rel = np.abs(lh - hist) / hist
    print("run", r, "psnr", round(orc.crop_psnr(img.abs().cpu(), gt), 3), "rel loss diff @it 0,1,2,4,8,16,29:",
          ["%.1e" % rel[i] for i in (0, 1, 2, 4, 8, 16, 29)], flush=True)
