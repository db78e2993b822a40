"""Error of the fp32 MLP kernels against a float64 evaluation of the same fp32 inputs: relative L2 and the SIGNED bias
(mean of (x - ref) * sign(ref) over mean |ref|) of output, d enc, dW1, dW2.  Run once per implementation
(IMMOCO_MLP_IMPL unset = matrix cores, =valu = VALU kernels); GPU box.
    python tools/diag_mlp_precision.py [n_points=1024000]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import miccai24_immoco_amd as pkg
from miccai24_immoco_amd import _lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024000
print("IMMOCO_MLP_IMPL =", os.environ.get("IMMOCO_MLP_IMPL"), "n =", n)
for which in ("image", "motion"):
    net = pkg.network_config if which == "image" else pkg.mot_network_config
    cfg = L.mlp_cfg(32, 2, net)
    hid, pad = cfg.n_hidden, cfg.n_out_padded
    nn = n // 10 if which == "image" else n
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(nn, 32, generator=g) * 0.3).cuda()
    w1 = (torch.randn(hid, 32, generator=g) * 0.2).cuda()
    w2 = (torch.randn(pad, hid, generator=g) * 0.2).cuda()
    w2[2:] = 0
    dout = (torch.randn(nn, 2, generator=g) * 0.05).cuda()
    xr, w1r, w2r = (t.double().requires_grad_(True) for t in (x, w1, w2))
    pre = xr @ w1r.t()
    h = torch.relu(pre) if which == "image" else torch.tanh(pre)
    out = (h @ w2r.t())[:, :2]
    (out * dout.double()).sum().backward()
    st = L.stream_ptr()
    o = torch.empty(nn, 2, device="cuda")
    L.check(L.lib().immoco_mlp_fwd(C.byref(cfg), L.ptr(x), 32, 2, nn, L.ptr(w1), L.ptr(w2), L.ptr(o), st))
    dx = torch.empty(nn, 32, device="cuda")
    dw1 = torch.zeros(hid, 32, device="cuda")
    dw2 = torch.zeros(pad, hid, device="cuda")
    L.check(L.lib().immoco_mlp_bwd(C.byref(cfg), L.ptr(x), 32, 2, nn, L.ptr(w1), L.ptr(w2), L.ptr(dout), L.ptr(dx),
                                   L.ptr(dw1), L.ptr(dw2), st))
    torch.cuda.synchronize()

    def rep(a, b, what):
        a, b = a.double(), b.double()
        rel = float((a - b).norm() / b.norm())
        bias = float(((a - b) * b.sign()).mean() / b.abs().mean())
        # torch's own fp32 evaluation for scale
        print(f"  {which:6s} {what:6s} rel L2 {rel:.2e}  signed bias {bias:+.2e}")
    rep(o, out.detach(), "out"); rep(dx, xr.grad, "d enc"); rep(dw1, w1r.grad, "dW1"); rep(dw2[:2], w2r.grad[:2], "dW2")
    # torch fp32 on the GPU (rocBLAS) for scale
    xf, w1f, w2f = (t.clone().requires_grad_(True) for t in (x, w1, w2))
    pre = xf @ w1f.t()
    h = torch.relu(pre) if which == "image" else torch.tanh(pre)
    of = (h @ w2f.t())[:, :2]
    (of * dout).sum().backward()
    print("  torch fp32 on the GPU:")
    rep(of.detach(), out.detach(), "out"); rep(xf.grad, xr.grad, "d enc"); rep(w1f.grad, w1r.grad, "dW1"); rep(w2f.grad[:2], w2r.grad[:2], "dW2")
