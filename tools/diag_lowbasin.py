"""When do HIP runs of C2 slice 1 leave the ~39.6 dB plateau?  N runs sampled every 25 iterations from 100 to 1400; for the
runs whose 600..1375 median is below 38 dB: the PSNR trace and the loss events (GPU box).
    python tools/diag_lowbasin.py [N=40] [--mlp-fp16 | --bf16x2] [--atomic] [--serial] [--seeds=K] [--dump=TAG]
--seeds=K: run r starts from init_params seed 2001 + r % K instead of the reference's fixed 1337 (per-seed counts printed)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import miccai24_immoco_amd as pkg
from miccai24_immoco_amd import synth
from miccai24_immoco_amd.models.immoco import get_solver
from miccai24_immoco_amd.utils.motion_utils import masks_to_col_group
from oracle import synth_cpu
from _stats import hip_psnr_samples
a = [x for x in sys.argv[1:] if not x.startswith("--")]
N = int(a[0]) if a else 40
s_ = synth_cpu.make_slice(320, 320, 10, 1)
k, lines = s_["kspace"].cuda(), s_["lines"].cuda()
masks = pkg.extract_movement_groups(lines, make_list=True)
gt = synth.phantom(320, 320, 1001).abs()
sol = get_solver(torch.device("cuda", 0), 320, 320, 10, atomic_scatter="--atomic" in sys.argv,
                 mlp_fp16=("bf16x2" if "--bf16x2" in sys.argv else "--mlp-fp16" in sys.argv),
                 serial_chains=(True if "--serial" in sys.argv else None))
kin, cg = k / k.abs().max() * 16000, masks_to_col_group(masks)
grid = list(range(100, 1400, 25))


def events(l, a, b, thr=1.5):
    ev, t = [], a
    while t < b:
        med = np.median(l[t - 20:t])
        if l[t] > thr * med:
            ev.append((t, round(float(l[t] / med), 1))); t += 40
        else:
            t += 1
    return ev


K = next((int(x.split("=")[1]) for x in sys.argv if x.startswith("--seeds=")), 0)
per_seed = {}
all_seeds, all_traces = [], []
low = 0
for r in range(N):
    seed = 2001 + r % K if K else 1337
    ps, loss = hip_psnr_samples(sol, kin, cg, gt, 3000, grid, seed=seed)
    tr = np.array([ps[t] for t in grid])
    plat = float(np.median(tr[[grid.index(t) for t in range(600, 1400, 25)]]))
    all_seeds.append(seed); all_traces.append(tr)
    c = per_seed.setdefault(seed, [0, 0, []])
    c[1] += 1
    c[2].append(round(plat, 2))
    if plat < 38.0:
        low += 1
        c[0] += 1
        first = next((t for t, v in zip(grid, tr) if t >= 300 and v < 37.5), None)
        print(f"run {r}: plateau {plat:.2f}; first sample < 37.5 dB at {first}; loss events {events(loss.astype(float), 100, 1375)}")
        print("   trace 100..1375:", np.round(tr, 1).tolist())
tag = next((x.split("=")[1] for x in sys.argv if x.startswith("--dump=")), None)
if tag:     # PSNR traces for offline comparison with oracle draws (gpurun_out/ travels back)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"lowbasin_{tag}.npz"), seed=np.array(all_seeds), grid=np.array(grid),
                        psnr=np.array(all_traces, dtype=np.float32))
if K:
    print("per seed (low, runs, plateau medians):", per_seed)
print("flags", [x for x in sys.argv if x.startswith("--")], "low-plateau runs:", low, "of", N)
