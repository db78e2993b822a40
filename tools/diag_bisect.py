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


import ctypes as C
from miccai24_immoco_amd import _lib as L
from miccai24_immoco_amd import tcnn as T


class _EncHIP(torch.autograd.Function):
    """hash-grid encode forward (lattice kernel) and backward (transposed-index plan) of the package, alone"""
    @staticmethod
    def forward(ctx, tab, mod):
        n = mod.n
        enc = torch.empty((16, n, 2), device=dev, dtype=torch.float32)
        pl = mod.plan
        L.check(L.lib().immoco_hashgrid_fwd_lattice(C.byref(mod.gcfg), pl.nM, pl.H, pl.W, L.ptr(pl.axes[0]), L.ptr(pl.axes[1]),
                                                    L.ptr(pl.axes[2]), L.ptr(tab), L.ptr(enc), 2, 2 * n, L.stream_ptr()), "fwd")
        ctx.mod = mod
        return enc.permute(1, 0, 2).reshape(n, 32)

    @staticmethod
    def backward(ctx, g):
        mod = ctx.mod
        n = mod.n
        denc = g.reshape(n, 16, 2).permute(1, 0, 2).contiguous()
        dtab = torch.zeros(mod.n_entries * 2, device=dev, dtype=torch.float32)
        L.check(L.lib().immoco_grid_plan_bwd(mod.plan.handle, L.ptr(denc), L.ptr(dtab), L.stream_ptr()), "bwd")
        return dtab, None


class _MlpHIP(torch.autograd.Function):
    """the package's MLP forward / backward kernels (exact fp32, matrix cores) on a point-major encoding"""
    @staticmethod
    def forward(ctx, enc, w1, w2, mod):
        n = enc.shape[0]
        enc = enc.contiguous()
        out = torch.empty((n, 2), device=dev, dtype=torch.float32)
        L.check(L.lib().immoco_mlp_fwd(C.byref(mod.mcfg), L.ptr(enc), 32, 2, n, L.ptr(w1.contiguous()), L.ptr(w2.contiguous()), L.ptr(out), L.stream_ptr()), "mlp_fwd")
        ctx.mod = mod
        ctx.save_for_backward(enc, w1, w2)
        return out

    @staticmethod
    def backward(ctx, dout):
        enc, w1, w2 = ctx.saved_tensors
        mod = ctx.mod
        n = enc.shape[0]
        denc = torch.empty_like(enc)
        dw1, dw2 = torch.zeros_like(w1), torch.zeros_like(w2)
        L.check(L.lib().immoco_mlp_bwd(C.byref(mod.mcfg), L.ptr(enc), 32, 2, n, L.ptr(w1.contiguous()), L.ptr(w2.contiguous()),
                                       L.ptr(dout.contiguous()), L.ptr(denc), L.ptr(dw1), L.ptr(dw2), L.stream_ptr()), "mlp_bwd")
        return denc, dw1, dw2, None


class _MlpMixed(torch.autograd.Function):
    """fwd_hip: the forward through the package's kernel and the backward through torch's formulas (or the other way round)"""
    @staticmethod
    def forward(ctx, enc, w1, w2, mod, fwd_hip):
        n = enc.shape[0]
        enc = enc.contiguous()
        act = torch.relu if mod.o.mlp.activation == "relu" else torch.tanh
        if fwd_hip:
            out = torch.empty((n, 2), device=dev, dtype=torch.float32)
            L.check(L.lib().immoco_mlp_fwd(C.byref(mod.mcfg), L.ptr(enc), 32, 2, n, L.ptr(w1.contiguous()), L.ptr(w2.contiguous()), L.ptr(out), L.stream_ptr()), "mlp_fwd")
        else:
            out = (act(enc @ w1.t()) @ w2.t())[:, :2].contiguous()
        ctx.mod, ctx.fwd_hip = mod, fwd_hip
        ctx.save_for_backward(enc, w1, w2)
        return out

    @staticmethod
    def backward(ctx, dout):
        enc, w1, w2 = ctx.saved_tensors
        mod = ctx.mod
        n = enc.shape[0]
        if ctx.fwd_hip:      # torch backward
            pre = enc @ w1.t()
            relu = mod.o.mlp.activation == "relu"
            h = torch.relu(pre) if relu else torch.tanh(pre)
            dpad = torch.zeros((n, w2.shape[0]), device=dev)
            dpad[:, :2] = dout
            dw2 = dpad.t() @ h
            dh = dpad @ w2
            dpre = dh * ((h > 0).float() if relu else (1 - h * h))
            return dpre @ w1, dpre.t() @ enc, dw2, None, None
        denc = torch.empty_like(enc)
        dw1, dw2 = torch.zeros_like(w1), torch.zeros_like(w2)
        L.check(L.lib().immoco_mlp_bwd(C.byref(mod.mcfg), L.ptr(enc), 32, 2, n, L.ptr(w1.contiguous()), L.ptr(w2.contiguous()),
                                       L.ptr(dout.contiguous()), L.ptr(denc), L.ptr(dw1), L.ptr(dw2), L.stream_ptr()), "mlp_bwd")
        return denc, dw1, dw2, None, None


class HybridINR(torch.nn.Module):
    """enc = "hip" | "torch", mlp = "hip" | "torch": one INR assembled from the package's kernels and the device oracle's
    torch expressions, same parameters / initialisation as both."""
    def __init__(self, dims, net_cfg, x, enc, mlp):
        super().__init__()
        self.o = orc.OracleINR(dims, 2, orc.encoding_config, net_cfg, seed=1337, device=dev)
        self.params = self.o.params
        self.enc_kind, self.mlp_kind, self.x = enc, mlp, x
        self.n = x.shape[0]
        self.gcfg = L.grid_cfg(dims, pkg.encoding_config)
        self.mcfg = L.mlp_cfg(32, 2, net_cfg)
        self.n_entries = self.o.geo.n_entries
        lat = T._detect_lattice(x)
        self.plan = T._GridPlan(self.gcfg, *lat)

    def forward(self, x):
        w1, w2, tab = self.o.split()
        enc = _EncHIP.apply(tab.reshape(-1), self) if self.enc_kind == "hip" else self.o.plan_for(x).encode(tab)
        if self.mlp_kind == "hip":
            return _MlpHIP.apply(enc, w1, w2, self)
        if self.mlp_kind in ("hipfwd", "hipbwd"):
            return _MlpMixed.apply(enc, w1, w2, self, self.mlp_kind == "hipfwd")
        pre = enc @ w1.t()
        h = torch.relu(pre) if self.o.mlp.activation == "relu" else torch.tanh(pre)
        return (h @ w2.t())[:, :2]


def make_inrs():
    if cfg["inr"].startswith("hyb"):      # inr=hyb:<image enc><image mlp><motion enc><motion mlp>, each h | t
        k = cfg["inr"].split(":")[1]
        m = {"h": "hip", "t": "torch", "f": "hipfwd", "b": "hipbwd"}
        return (HybridINR(2, orc.network_config, identy.view(-1, 2), m[k[0]], m[k[1]]),
                HybridINR(3, orc.mot_network_config, grid_in, m[k[2]], m[k[3]]))
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
