"""Which component moves the 200-iteration PSNR level of a slice?  The reference's loop (immoco.py:164-181) driven from
Python with torch.optim.Adam, every operator either HIP (the package's op-level modules) or torch (the device oracle's):
    python tools/diag_bisect.py <slice> <runs> inr=hip|torch warp=hip|torch fft=hip|torch ge=hip|torch sel=hip|torch
Prints the median-of-last-21 PSNR over the runs (GPU box; diagnostics)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch, torch.nn.functional as F
import miccai24_immoco_amd as pkg
from miccai24_immoco_amd.models import immoco as M
from oracle import immoco_oracle as orc
from device_oracle_sampler import slice_input, device_psnr
sl, runs = int(sys.argv[1]), int(sys.argv[2])
cfg = dict(inr="hip", warp="hip", fft="hip", ge="hip", sel="hip", plans="1")   # plans=0: generic kernels + atomic scatter
cfg.update(dict(a.split("=") for a in sys.argv[3:]))
dev = torch.device("cuda", 0)
k, lines, gt = slice_input(sl)
masks_c = orc.extract_movement_groups(lines, make_list=True)
masks = masks_c.to(dev)
nM, H, W = masks.shape
kin = (k / k.abs().max() * 16000).to(dev)
gt_d = gt.to(dev)
lam = orc.lambda_schedule(200, 1e-2)
identy = orc.identity_grid(H, W).to(dev)
grid_in = orc.make_grids((nM, H, W)).to(dev)
cg = M.masks_to_col_group(masks)
ge_hip = pkg.GradientEntropyLoss()


def make_inrs():
    if cfg["inr"] in ("hip_img", "hip_mot"):      # one INR from the package, the other from the device oracle
        lp = cfg["plans"] == "1"
        hi = pkg.NetworkWithInputEncoding(2, 2, pkg.encoding_config, pkg.network_config, seed=1337, device=dev, lattice_plans=lp)
        hm = pkg.NetworkWithInputEncoding(3, 2, pkg.encoding_config, pkg.mot_network_config, seed=1337, device=dev, lattice_plans=lp)
        ti = orc.OracleINR(2, 2, orc.encoding_config, orc.network_config, seed=1337, device=dev)
        tm = orc.OracleINR(3, 2, orc.encoding_config, orc.mot_network_config, seed=1337, device=dev)
        return (hi, tm) if cfg["inr"] == "hip_img" else (ti, hm)
    if cfg["inr"] == "hip":
        lp = cfg["plans"] == "1"
        return (pkg.NetworkWithInputEncoding(2, 2, pkg.encoding_config, pkg.network_config, seed=1337, device=dev, lattice_plans=lp),
                pkg.NetworkWithInputEncoding(3, 2, pkg.encoding_config, pkg.mot_network_config, seed=1337, device=dev, lattice_plans=lp))
    return (orc.OracleINR(2, 2, orc.encoding_config, orc.network_config, seed=1337, device=dev),
            orc.OracleINR(3, 2, orc.encoding_config, orc.mot_network_config, seed=1337, device=dev))


def forward(img_inr, mot_inr):
    o = img_inr(identy.view(-1, 2)).float().view(H, W, 2)
    ip = torch.view_as_complex(o.contiguous())
    grids = mot_inr(grid_in).float().tanh().view(nM, H, W, 2) + identy.view(1, H, W, 2)
    if cfg["warp"] == "hip":
        mi = M._Warp.apply(ip, grids)
    else:
        images = ip.unsqueeze(0).repeat(nM, 1, 1)
        mi = torch.view_as_complex(F.grid_sample(torch.view_as_real(images).permute(0, 3, 1, 2), grids, mode="bilinear",
                                                 align_corners=False, padding_mode="zeros").permute(0, 2, 3, 1).contiguous())
    fft = pkg.FFT if cfg["fft"] == "hip" else orc.FFT
    if cfg["sel"] == "hip":
        kout = M._LineSelect.apply(fft(torch.cat([ip.unsqueeze(0), mi], dim=0)), cg)
    else:
        kout = fft(ip) * (1 - masks.sum(0)).float() + (fft(mi) * masks.float()).sum(0)
    return kout, ip


stats = []
for r in range(runs):
    img_inr, mot_inr = make_inrs()
    opt = torch.optim.Adam([{"params": mot_inr.parameters(), "lr": 1e-2}, {"params": img_inr.parameters(), "lr": 1e-2}])
    ps = torch.zeros(200, device=dev)
    for j in range(200):
        opt.zero_grad()
        kf, ip = forward(img_inr, mot_inr)
        ge = ge_hip(ip) if cfg["ge"] == "hip" else orc.gradient_entropy_loss(ip)
        loss = F.mse_loss(torch.view_as_real(kf), torch.view_as_real(kin)) + ge * lam[j]
        loss.backward()
        opt.step()
        with torch.no_grad():
            ps[j] = device_psnr(ip.detach().abs(), gt_d)
    stats.append(float(ps[179:200].median()))
st = np.array(stats)
print("slice", sl, cfg, f"{runs} runs: median-of-last-21 PSNR mean {st.mean():.3f} sd {st.std(ddof=1):.3f} se {st.std(ddof=1) / np.sqrt(len(st)):.3f}", flush=True)
