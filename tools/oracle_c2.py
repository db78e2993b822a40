"""Run the CPU oracle on config C2 (320x320, 10 groups) for N iterations; log loss/PSNR; save npz.
usage: python tools/oracle_c2.py <slice_idx> <iters> <out.npz>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, torch.nn.functional as F
from oracle import immoco_oracle as orc
from miccai24_immoco_amd import synth
idx, iters, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
torch.set_num_threads(int(os.environ.get("ORACLE_THREADS", "8")))
s = synth.make_slice(320, 320, 10, idx)
masks = orc.extract_movement_groups(s["lines"], make_list=True)
gt = s["gt"].abs()
model = orc.OracleIMMoCo(masks, image_inr=orc.OracleINR(2, 2, orc.encoding_config, orc.network_config),
                         motion_inr=orc.OracleINR(3, 2, orc.encoding_config, orc.mot_network_config))
k = s["kspace"]
kin = k.div(k.abs().max()).mul(16000).clone()
opt = torch.optim.Adam([{"params": model.motion_inr.parameters(), "lr": 1e-2}, {"params": model.image_inr.parameters(), "lr": 1e-2}])
lam = orc.lambda_schedule(iters, 1e-2)
hist, psnrs = [], {}
t0 = time.time()
for j in range(iters):
    opt.zero_grad()
    kf, ip = model()
    loss = F.mse_loss(torch.view_as_real(kf), torch.view_as_real(kin)) + orc.gradient_entropy_loss(ip) * lam[j]
    loss.backward()
    opt.step()
    hist.append(float(loss))
    if j % 25 == 0 or j == iters - 1:
        psnrs[j] = orc.crop_psnr(ip.detach().abs(), gt)
        print(j, f"loss {hist[-1]:.3f} psnr {psnrs[j]:.3f} t={time.time()-t0:.0f}s", flush=True)
        np.savez_compressed(out, image=ip.detach().numpy(), kfm=kf.detach().numpy(), loss=np.array(hist, dtype=np.float64),
                            psnr_iters=np.array(list(psnrs.keys())), psnr=np.array(list(psnrs.values())), slice_idx=idx, iters_done=j + 1)
