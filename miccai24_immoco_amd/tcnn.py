"""`tinycudann.NetworkWithInputEncoding`-compatible module on the HIP kernels.

Mirrors the third-party operator the reference path sits behind
(reference src/models/immoco.py:1,60-65,85,93): same constructor arguments, a
single flat fp32 ``params`` Parameter ([W1 | W2(padded) | hash table], tcnn
order), ``forward(x[N, n_in]) -> [N, n_out]``.  Differences, by design: compute
and output are fp32 (tcnn: fp16 with loss scale 128), and the initial values
come from the library's counter-based generator instead of tcnn's RNG stream.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L


class _INRFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, params, mod):
        L.require_gpu(x, params, what="NetworkWithInputEncoding")
        x = x.contiguous().float()
        n = x.shape[0]
        w1, w2, tab = mod._split(params)
        enc = torch.empty((n, 32), device=x.device, dtype=torch.float32)
        out = torch.empty((n, 2), device=x.device, dtype=torch.float32)
        lib, st = L.lib(), L.stream_ptr()
        L.check(lib.immoco_hashgrid_fwd(C.byref(mod.grid_cfg), L.ptr(x), n, L.ptr(tab), L.ptr(enc), 32, 2, st),
                "hashgrid_fwd")
        L.check(lib.immoco_mlp_fwd(C.byref(mod.mlp_cfg), L.ptr(enc), 32, 2, n, L.ptr(w1), L.ptr(w2), L.ptr(out), st),
                "mlp_fwd")
        ctx.mod = mod
        ctx.save_for_backward(x, params, enc)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, params, enc = ctx.saved_tensors
        mod = ctx.mod
        n = x.shape[0]
        dout = dout.contiguous().float()
        w1, w2, tab = mod._split(params)
        dparams = torch.zeros_like(params)
        dw1, dw2, dtab = mod._split(dparams)
        denc = torch.empty_like(enc)
        lib, st = L.lib(), L.stream_ptr()
        L.check(lib.immoco_mlp_bwd(C.byref(mod.mlp_cfg), L.ptr(enc), 32, 2, n, L.ptr(w1), L.ptr(w2), L.ptr(dout),
                                   L.ptr(denc), L.ptr(dw1), L.ptr(dw2), st), "mlp_bwd")
        L.check(lib.immoco_hashgrid_bwd(C.byref(mod.grid_cfg), L.ptr(x), n, L.ptr(denc), 32, 2, L.ptr(dtab), st),
                "hashgrid_bwd")
        return None, dparams, None


class NetworkWithInputEncoding(torch.nn.Module):
    def __init__(self, n_input_dims, n_output_dims, encoding_config, network_config, seed=1337, device="cuda"):
        super().__init__()
        if n_output_dims != 2:
            raise L.ImmocoError("only n_output_dims=2 is supported (the reference's INRs)")
        self.n_input_dims, self.n_output_dims, self.seed = n_input_dims, n_output_dims, seed
        self.grid_cfg = L.grid_cfg(n_input_dims, encoding_config)
        self.mlp_cfg = L.mlp_cfg(self.grid_cfg.n_levels * self.grid_cfg.n_features, n_output_dims, network_config)
        geo = L.geometry(self.grid_cfg)
        self.n_entries = int(geo.offset[self.grid_cfg.n_levels])
        self.n_w1 = self.mlp_cfg.n_hidden * self.mlp_cfg.n_in
        self.n_w2 = self.mlp_cfg.n_out_padded * self.mlp_cfg.n_hidden
        n_params = self.n_w1 + self.n_w2 + 2 * self.n_entries
        dev = torch.device(device)
        if dev.type != "cuda":
            raise L.ImmocoError("NetworkWithInputEncoding lives on the GPU only (like tiny-cuda-nn)")
        p = torch.empty(n_params, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            L.check(L.lib().immoco_init_params(C.byref(self.grid_cfg), C.byref(self.mlp_cfg), seed & 0xFFFFFFFF,
                                               L.ptr(p), L.stream_ptr()), "init_params")
        self.params = torch.nn.Parameter(p)

    def _split(self, p):
        a, b = self.n_w1, self.n_w1 + self.n_w2
        return p[:a], p[a:b], p[b:]

    def forward(self, x):
        return _INRFunction.apply(x, self.params, self)
