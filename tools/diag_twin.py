"""Diagnostic: transposed-index backward (twin entries on/off via IMMOCO_CSR_NO_TWIN) vs the atomic scatter
on the C2 lattice, and vs float64 accumulation of the atomic kernel's own terms."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import miccai24_immoco_amd as pkg
from miccai24_immoco_amd import _lib as L
from miccai24_immoco_amd.models.immoco import make_grids

nM, H, W = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (10, 320, 320)))
n = nM * H * W
cfg = L.grid_cfg(3, pkg.encoding_config)
geo = L.GridGeometry() if hasattr(L, "GridGeometry") else None
ax = [torch.linspace(-1, 1, k).cuda() if k > 1 else torch.tensor([-1.0]).cuda() for k in (nM, H, W)]
coords = make_grids((nM, H, W)).cuda().contiguous()
g = torch.Generator(device="cuda").manual_seed(1)
d_lm = torch.randn(16, n, 2, device="cuda", generator=g)
st = L.stream_ptr()
geom = L.query_geometry(cfg) if hasattr(L, "query_geometry") else None
n_entries = 7114752
plan = C.c_void_p()
L.check(L.lib().immoco_grid_plan_create(C.byref(cfg), nM, H, W, L.ptr(ax[0]), L.ptr(ax[1]), L.ptr(ax[2]),
                                        C.byref(plan), st), "plan")
print("plan bytes", L.lib().immoco_grid_plan_bytes(plan), "full entries x8", n * 16 * 8 * 8)
dt = torch.zeros(n_entries, 2, device="cuda")
L.check(L.lib().immoco_grid_plan_bwd(plan, L.ptr(d_lm), L.ptr(dt), st), "bwd")
da = torch.zeros(n_entries, 2, device="cuda")
L.check(L.lib().immoco_hashgrid_bwd(C.byref(cfg), L.ptr(coords), n, L.ptr(d_lm), 2, 2 * n, L.ptr(da), st))
torch.cuda.synchronize()
s = da.abs().max().item()
err = (dt - da).abs()
print("max |plan - atomic| / max", err.max().item() / s, "mean", err.mean().item() / s, "n > 1e-4*s:", int((err > 1e-4 * s).sum()))
bad = (err > 1e-4 * s).nonzero()
print("first bad", bad[:10].tolist())
