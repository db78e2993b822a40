"""Rare-corruption hunt (GPU box): ONE Adam iteration repeated R times from the SAME late state (parameters + moments
after K iterations of a C2 solve).  Every repetition evaluates the same function; the only legitimate difference is
the fp32 summation order of the atomics (~1e-6 relative in the gradient).  A race or a stale buffer shows up as an
outlier.     python tools/diag_repeat_step.py [K=300] [R=400] [--mlp-fp16] [--table-fp16] [--no-graph]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import miccai24_immoco_amd as pkg
from miccai24_immoco_amd import synth
from miccai24_immoco_amd.models.immoco import get_solver, lambda_schedule
from miccai24_immoco_amd.utils.motion_utils import masks_to_col_group
args = [a for a in sys.argv[1:] if not a.startswith("--")]
K = int(args[0]) if args else 300
R = int(args[1]) if len(args) > 1 else 400
dev = torch.device("cuda", 0)
s = synth.make_slice(320, 320, 10, 1, device=dev)
masks = pkg.extract_movement_groups(s["lines"], make_list=True)
sol = get_solver(dev, 320, 320, int(masks.shape[0]), use_graph="--no-graph" not in sys.argv,
                 table_fp16="--table-fp16" in sys.argv, mlp_fp16="--mlp-fp16" in sys.argv)
k = s["kspace"]
kin = k / k.abs().max() * 16000
cg = masks_to_col_group(masks)
lam = lambda_schedule(3000, 1e-2)
pi, pm = sol.init_params()
ai, am = torch.zeros(2 * pi.numel(), device=dev), torch.zeros(2 * pm.numel(), device=dev)
sol.solve(kin, cg, pi, pm, ai, am, K, 1e-2, lam[:K])
state = [t.clone() for t in (pi, pm, ai, am)]
ni, nm = pi.numel(), pm.numel()
ref = None
rows = []
for r in range(R):
    p_i, p_m, a_i, a_m = [t.clone() for t in state]
    img, _, loss = sol.solve(kin, cg, p_i, p_m, a_i, a_m, 1, 1e-2, lam[K:K + 1], step0=K, want_loss=True)
    gi = (a_i[:ni] - 0.9 * state[2][:ni]) / 0.1
    gm = (a_m[:nm] - 0.9 * state[3][:nm]) / 0.1
    if ref is None:
        ref = (gi.clone(), gm.clone(), img.clone(), float(loss[0]))
        continue
    rows.append((float((gi - ref[0]).norm() / ref[0].norm()), float((gm - ref[1]).norm() / ref[1].norm()),
                 float((gi - ref[0]).abs().max() / ref[0].abs().max()), float((gm - ref[1]).abs().max() / ref[1].abs().max()),
                 float((img - ref[2]).abs().max() / ref[2].abs().max()), abs(float(loss[0]) - ref[3]) / abs(ref[3])))
a = np.array(rows)
names = ["image grad rel L2", "motion grad rel L2", "image grad max/max", "motion grad max/max", "forward image max/max", "loss rel"]
print(f"K={K} R={R} flags={[x for x in sys.argv if x.startswith('--')]}")
for j, nme in enumerate(names):
    c = a[:, j]
    print(f"{nme:24s} median {np.median(c):.3e}  p99 {np.quantile(c, 0.99):.3e}  max {c.max():.3e}  (rep {int(c.argmax()) + 1})")
bad = np.where((a[:, 0] > 20 * np.median(a[:, 0])) | (a[:, 1] > 20 * np.median(a[:, 1])) | (a[:, 4] > 1e-4))[0]
print("outliers (> 20x the median gradient distance, or a forward image off by > 1e-4):", [(int(b) + 1, a[b].tolist()) for b in bad[:10]], "count", len(bad))
