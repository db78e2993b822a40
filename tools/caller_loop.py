"""The reference's OWN loop (src/models/immoco.py:164-175: zero_grad / model() / mse_loss + GE / backward /
torch.optim.Adam.step) driven from Python on the module API (IMMoCo + NetworkWithInputEncoding + FFT +
GradientEntropyLoss), timed per iteration at config C2 next to the fused solver.  GPU box.
    python tools/caller_loop.py [--iters 100]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
import miccai24_immoco_amd as pkg
from miccai24_immoco_amd import synth
ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=100)
ap.add_argument("--fused-adam", action="store_true", help="torch.optim.Adam(fused=True)")
a = ap.parse_args()
dev = torch.device("cuda", 0)
s = synth.make_slice(320, 320, 10, 1, device=dev)
masks = pkg.extract_movement_groups(s["lines"], make_list=True)
k = s["kspace"]
kin = (k / k.abs().max() * 16000).detach()
model = pkg.IMMoCo(masks)
opt = torch.optim.Adam([{"params": model.motion_inr.parameters(), "lr": 1e-2},
                        {"params": model.image_inr.parameters(), "lr": 1e-2}], fused=a.fused_adam)
ge = pkg.GradientEntropyLoss()
lam = 1e-2


def one():
    opt.zero_grad()
    kf, ip = model()
    loss = F.mse_loss(torch.view_as_real(kf), torch.view_as_real(kin)) + ge(ip).mul(lam)
    loss.backward()
    opt.step()
    return loss


for _ in range(5):
    one()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    l = one()
torch.cuda.synchronize()
ms_loop = (time.perf_counter() - t0) / a.iters * 1e3
pkg.imcoco_motion_correction(k, masks, iters=a.iters)
torch.cuda.synchronize()
t0 = time.perf_counter()
pkg.imcoco_motion_correction(k, masks, iters=a.iters)
torch.cuda.synchronize()
ms_fused = (time.perf_counter() - t0) / a.iters * 1e3
print(f"caller-driven loop (module API + torch.optim.Adam{' fused' if a.fused_adam else ''}): {ms_loop:.3f} ms/iteration; "
      f"fused solver: {ms_fused:.3f} ms/iteration; ratio {ms_loop / ms_fused:.2f}; final loss {float(l):.3f}")
