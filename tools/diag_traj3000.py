"""Diagnostic: PSNR / loss trajectory of a 3000-iteration HIP solve on C2 slice 1, in segments."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import miccai24_immoco_amd as pkg
from oracle import synth_cpu as synth
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
# The oracle (tools/oracle_c2.py) logs loss and PSNR of the forward pass of iterations 0, 25, 50, ...: solve in
# segments that END at those iterations (the solver returns the tensors of a segment's last forward pass).
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = []
for rep in range(reps):
    pi, pm = sol.init_params()
    ai.zero_(); am.zero_()
    a, out = 0, []
    for end in [0] + list(range(25, 3000, 25)) + [2999]:
        n = end - a + 1
        img, kf, loss = sol.solve(kin, cg, pi, pm, ai, am, n, 1e-2, lam[a:a + n], step0=a, want_loss=True)
        out.append((end, float(loss[-1]), crop_psnr(img.abs().cpu(), s["gt"].abs())))
        a = end + 1
    rows.append(out)
    print("rep", rep, " ".join(f"{e}:{l:.4g}/{p:.2f}" for e, l, p in out[::8]), flush=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"traj3000_hip_slice{idx}.npz"),
                    iters=np.array([e for e, _, _ in rows[0]]), loss=np.array([[l for _, l, _ in r] for r in rows]),
                    psnr=np.array([[p for _, _, p in r] for r in rows]))
