"""CPU-oracle loss record at BASELINE config 5's own shape AND precision: 640x640, 20 motion groups, fp16 hash-grid
features (OracleINR(table_fp16=True): gather from the fp16-rounded table, fp32 master + straight-through gradient),
first N iterations of the 3000-iteration schedule -> tests/golden/c5_oracle_fp16.npz.  Build container (~10 GB).
    python tools/oracle_c5.py [iters=5] [threads=4]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import immoco_oracle as orc, synth_cpu
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
torch.set_num_threads(int(sys.argv[2]) if len(sys.argv) > 2 else 4)
H, W, NM, IDX = 640, 640, 20, 7
s = synth_cpu.make_slice(H, W, NM, IDX)
masks = orc.extract_movement_groups(s["lines"], make_list=True)
model = orc.OracleIMMoCo(masks, image_inr=orc.OracleINR(2, 2, orc.encoding_config, orc.network_config, table_fp16=True),
                         motion_inr=orc.OracleINR(3, 2, orc.encoding_config, orc.mot_network_config, table_fp16=True))
hist = []
t0 = time.time()
# the schedule of the full 3000-iteration solve: lambda_GE = 1e-2 throughout the first iterations
lam3000 = orc.lambda_schedule(3000, 1e-2)[:iters]
assert all(v == 1e-2 for v in lam3000)
import torch.nn.functional as F
k = s["kspace"]
kin = k.div(k.abs().max()).mul(16000).clone()
opt = torch.optim.Adam([{"params": model.motion_inr.parameters(), "lr": 1e-2}, {"params": model.image_inr.parameters(), "lr": 1e-2}])
for j in range(iters):
    opt.zero_grad()
    kf, ip = model()
    loss = F.mse_loss(torch.view_as_real(kf), torch.view_as_real(kin)) + orc.gradient_entropy_loss(ip) * lam3000[j]
    loss.backward()
    opt.step()
    hist.append(float(loss.detach()))
    print(j, hist[-1], f"t={time.time() - t0:.0f}s", flush=True)
out = os.path.join(ROOT, "tests", "golden", "c5_oracle_fp16.npz")
np.savez_compressed(out, loss=np.array(hist, dtype=np.float64), H=H, W=W, n_movements=NM, slice_idx=IDX,
                    n_groups=int(masks.shape[0]), kspace_abs_sum=float(k.abs().double().sum()),
                    kspace_abs_max=float(k.abs().max()))
print(out, os.path.getsize(out), "bytes")
