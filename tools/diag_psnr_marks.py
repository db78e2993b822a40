import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import miccai24_immoco_amd as pkg
from miccai24_immoco_amd import synth
from miccai24_immoco_amd.models.immoco import get_solver, lambda_schedule
from miccai24_immoco_amd.utils.evaluate import crop_psnr
from miccai24_immoco_amd.utils.motion_utils import masks_to_col_group
from oracle import synth_cpu
sl, n = int(sys.argv[1]), int(sys.argv[2])
s_ = synth_cpu.make_slice(320, 320, 10, sl)
k, lines = s_["kspace"].cuda(), s_["lines"].cuda()
masks = pkg.extract_movement_groups(lines, make_list=True)
gt = synth.phantom(320, 320, 1000 + sl).abs()
sol = get_solver(torch.device("cuda", 0), 320, 320, int(masks.shape[0]))
kin = k / k.abs().max() * 16000
cg = masks_to_col_group(masks)
lam = lambda_schedule(3000, 1e-2)
marks = [100, 200, 400, 800, 1400]
for r in range(n):
    pi, pm = sol.init_params()
    ai, am = torch.zeros(2 * pi.numel(), device="cuda"), torch.zeros(2 * pm.numel(), device="cuda")
    a, row = 0, []
    for end in marks:
        m = end - a + 1
        img, _, _ = sol.solve(kin, cg, pi, pm, ai, am, m, 1e-2, lam[a:a + m], step0=a)
        row.append(crop_psnr(img.abs().cpu(), gt))
        a = end + 1
    print(r, " ".join("%d:%.2f" % (mk, p) for mk, p in zip(marks, row)), flush=True)
