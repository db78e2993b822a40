"""Diagnostic: forward-encode cost of each level of the motion grid alone (one-level grids with the same
geometry: base_resolution = 16 << l), C2 lattice."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import miccai24_immoco_amd as pkg
from miccai24_immoco_amd import _lib as L
from miccai24_immoco_amd.models.immoco import make_grids
nM, H, W = 10, 320, 320
n = nM * H * W
coords = make_grids((nM, H, W)).cuda().contiguous()
st = L.stream_ptr()
tot = 0.0
for l in range(16):
    enc = dict(pkg.encoding_config)
    enc["n_levels"] = 1
    enc["base_resolution"] = 16 << l
    cfg = L.grid_cfg(3, enc)
    geo = L.GridGeometry()
    L.check(L.lib().immoco_grid_geometry_query(C.byref(cfg), C.byref(geo)))
    ne = int(geo.offset[1])
    table = torch.rand(ne, 2, device="cuda")
    out = torch.empty(n, 2, device="cuda")
    for _ in range(2):
        L.check(L.lib().immoco_hashgrid_fwd(C.byref(cfg), L.ptr(coords), n, L.ptr(table), L.ptr(out), 2, 2 * n, st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        L.check(L.lib().immoco_hashgrid_fwd(C.byref(cfg), L.ptr(coords), n, L.ptr(table), L.ptr(out), 2, 2 * n, st))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    tot += ms
    print(f"level {l:2d} res {16 << l:7d} entries {ne:8d} hashed {int(geo.hashed[0])} ms {ms:.4f}", flush=True)
print("sum", round(tot, 4))
