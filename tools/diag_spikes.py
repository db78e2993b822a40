"""Every-iteration PSNR of HIP solves over a window (GPU box): shape of the excursions / single-iteration spikes.
    python tools/diag_spikes.py [runs=4] [t0=1300] [t1=1500] [--mlp-fp16]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import miccai24_immoco_amd as pkg
from miccai24_immoco_amd import synth
from miccai24_immoco_amd.models.immoco import get_solver, lambda_schedule
from miccai24_immoco_amd.utils.evaluate import crop_psnr
from miccai24_immoco_amd.utils.motion_utils import masks_to_col_group
from oracle import synth_cpu
a = [x for x in sys.argv[1:] if not x.startswith("--")]
runs, t0, t1 = (int(a[0]) if a else 4), (int(a[1]) if len(a) > 1 else 1300), (int(a[2]) if len(a) > 2 else 1500)
s_ = synth_cpu.make_slice(320, 320, 10, 1)
k, lines = s_["kspace"].cuda(), s_["lines"].cuda()
masks = pkg.extract_movement_groups(lines, make_list=True)
gt = synth.phantom(320, 320, 1001).abs()
sol = get_solver(torch.device("cuda", 0), 320, 320, 10, mlp_fp16="--mlp-fp16" in sys.argv)
kin = k / k.abs().max() * 16000
cg = masks_to_col_group(masks)
lam = lambda_schedule(3000, 1e-2)
for r in range(runs):
    pi, pm = sol.init_params()
    ai, am = torch.zeros(2 * pi.numel(), device="cuda"), torch.zeros(2 * pm.numel(), device="cuda")
    sol.solve(kin, cg, pi, pm, ai, am, t0, 1e-2, lam[:t0])
    ps, ls, mx = [], [], []
    for t in range(t0, t1):
        img, _, l = sol.solve(kin, cg, pi, pm, ai, am, 1, 1e-2, lam[t:t + 1], step0=t, want_loss=True)
        ab = img.abs()
        ps.append(crop_psnr(ab.cpu(), gt)); ls.append(float(l[0])); mx.append(float(ab[80:240, 80:240].max()))
    ps = np.array(ps)
    d = np.abs(np.diff(ps))
    print(f"run {r}: PSNR min {ps.min():.2f} max {ps.max():.2f} median {np.median(ps):.2f}; largest single-iteration jump {d.max():.2f} dB at {t0 + int(d.argmax())}; "
          f"crop max |img| min/max over window {min(mx):.1f}/{max(mx):.1f}")
    j = int(d.argmax())
    lo, hi = max(0, j - 6), min(len(ps), j + 8)
    print("   around the jump: psnr", np.round(ps[lo:hi], 2), "loss", np.round(ls[lo:hi], 3), "cropmax", np.round(mx[lo:hi], 1))
