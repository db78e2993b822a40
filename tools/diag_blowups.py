"""Loss blow-up events (loss > 1.5 x the median of the previous 20 iterations) of N HIP solves of C2 slice 1, iterations
300 .. 1500 of the 3000-iteration schedule, next to the same statistic of the CPU oracle's records (GPU box).
    python tools/diag_blowups.py [N=24] [--mlp-fp16] [--table-fp16]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import miccai24_immoco_amd as pkg
from miccai24_immoco_amd.models.immoco import get_solver, lambda_schedule
from miccai24_immoco_amd.utils.motion_utils import masks_to_col_group
from oracle import synth_cpu
a = [x for x in sys.argv[1:] if not x.startswith("--")]
N = int(a[0]) if a else 24


def events(l, a, b, thr=1.5):
    ev, t = [], a
    while t < b:
        med = np.median(l[t - 20:t])
        if l[t] > thr * med:
            ev.append((t, round(float(l[t] / med), 1)))
            t += 40
        else:
            t += 1
    return ev


s_ = synth_cpu.make_slice(320, 320, 10, 1)
k, lines = s_["kspace"].cuda(), s_["lines"].cuda()
masks = pkg.extract_movement_groups(lines, make_list=True)
sol = get_solver(torch.device("cuda", 0), 320, 320, 10, mlp_fp16="--mlp-fp16" in sys.argv, table_fp16="--table-fp16" in sys.argv)
kin = k / k.abs().max() * 16000
cg = masks_to_col_group(masks)
lam = lambda_schedule(3000, 1e-2)
firsts, counts = [], []
for r in range(N):
    pi, pm = sol.init_params()
    ai, am = torch.zeros(2 * pi.numel(), device="cuda"), torch.zeros(2 * pm.numel(), device="cuda")
    _, _, l = sol.solve(kin, cg, pi, pm, ai, am, 1500, 1e-2, lam[:1500], want_loss=True)
    e = events(l.cpu().numpy().astype(float), 300, 1500)
    firsts.append(e[0][0] if e else 1500)
    counts.append(len(e))
    print(r, e, flush=True)
print("HIP flags", [x for x in sys.argv if x.startswith("--")], ": events per run mean %.2f; first event: median %d, quartiles %s; runs without event %d/%d; runs with an event before 1100: %d"
      % (np.mean(counts), np.median(firsts), np.quantile(firsts, [.25, .75]).astype(int).tolist(), sum(c == 0 for c in counts), N, sum(f < 1100 for f in firsts)))
rec = np.load(os.path.join(ROOT, "tests", "golden", "c2_oracle_slice1_3000it.npz"))
for r, l in enumerate(rec["oracle_loss"].astype(float)):
    print("oracle record", r, events(l, 300, 1500))
