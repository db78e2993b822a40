"""IM-MoCo solver with the reference's names, arguments and return values
(reference src/models/immoco.py), computing on MI355X through libimmoco_hip.so.

* ``IMMoCo(masks)`` / ``IMMoCo.forward()`` — the forward k-space model
  (immoco.py:56-113) as autograd-capable HIP operators, so a caller can still
  drive it with its own optimiser loop exactly like the reference does.
* ``imcoco_motion_correction(kspace_corr, masks, iters, learning_rate,
  lambda_ge, debug)`` — the whole inner optimisation loop (immoco.py:116-206)
  executed by the fused native solver (one captured hipGraph per iteration, no
  host synchronisation inside the loop).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib as L
from ..tcnn import NetworkWithInputEncoding
from ..utils.data_utils import FFT
from ..utils.motion_utils import masks_to_col_group

# reference immoco.py:11-37 (same keys; "fine_resolution" is ignored, as upstream does)
network_config = {
    "otype": "CutLassMLP",
    "activation": "ReLU",
    "output_activation": "None",
    "n_neurons": 256,
    "n_hidden_layers": 1,
}

mot_network_config = {
    "otype": "FullyFusedMLP",
    "activation": "Tanh",
    "output_activation": "None",
    "n_neurons": 64,
    "n_hidden_layers": 1,
}

encoding_config = {
    "otype": "Grid",
    "type": "Hash",
    "n_levels": 16,
    "n_features_per_level": 2,
    "log2_hashmap_size": 19,
    "base_resolution": 16,
    "fine_resolution": 320,
    "per_level_scale": 2,
    "interpolation": "Linear",
}


def make_grids(sizes, device="cpu"):
    """immoco.py:48-53 — [prod(sizes), len(sizes)] lattice of linspace(-1, 1, s)."""
    lin = [torch.linspace(-1, 1, s, device=device) for s in sizes]
    mesh = torch.meshgrid(*lin, indexing="ij")
    return torch.stack(mesh, dim=-1).view(-1, len(sizes))


class _Warp(torch.autograd.Function):
    """grid_sample(bilinear, zeros, align_corners=False) of one complex image (immoco.py:91,97-107)."""

    @staticmethod
    def forward(ctx, image, grids):
        L.require_gpu(image, grids, what="warp")
        image = image.contiguous()
        grids = grids.contiguous().float()
        nM, H, W, _ = grids.shape
        out = torch.empty((nM, H, W), device=image.device, dtype=torch.complex64)
        L.check(L.lib().immoco_warp_fwd(L.ptr(image), L.ptr(grids), nM, H, W, L.ptr(out), L.stream_ptr()), "warp_fwd")
        ctx.save_for_backward(image, grids)
        return out

    @staticmethod
    def backward(ctx, dout):
        image, grids = ctx.saved_tensors
        nM, H, W, _ = grids.shape
        dout = dout.contiguous()
        dimage = torch.zeros_like(image)
        dgrids = torch.empty_like(grids)
        L.check(L.lib().immoco_warp_bwd(L.ptr(image), L.ptr(grids), L.ptr(dout), nM, H, W, L.ptr(dimage),
                                        L.ptr(dgrids), L.stream_ptr()), "warp_bwd")
        return dimage, dgrids


class _LineSelect(torch.autograd.Function):
    """K[:, c] = K_all[g(c)][:, c]  ==  FFT(img)*(1-sum masks) + sum(FFT(warp)*masks)  (immoco.py:109-111)."""

    @staticmethod
    def forward(ctx, kall, col_group):
        L.require_gpu(kall, col_group, what="line_select")
        kall = kall.contiguous()
        nM1, H, W = kall.shape
        out = torch.empty((H, W), device=kall.device, dtype=torch.complex64)
        L.check(L.lib().immoco_kspace_select(L.ptr(kall), L.ptr(col_group), nM1 - 1, H, W, L.ptr(out), L.stream_ptr()),
                "kspace_select")
        ctx.save_for_backward(col_group)
        ctx.nM1 = nM1
        return out

    @staticmethod
    def backward(ctx, g):
        (col_group,) = ctx.saved_tensors
        sel = torch.arange(ctx.nM1, device=g.device, dtype=torch.int32).view(-1, 1, 1) == col_group.view(1, 1, -1)
        return g.unsqueeze(0) * sel, None


class IMMoCo(nn.Module):
    def __init__(self, masks, seed=1337):
        super().__init__()
        L.require_gpu(masks, what="IMMoCo(masks)")
        dev = masks.device
        self.image_inr = NetworkWithInputEncoding(2, 2, encoding_config, network_config, seed=seed, device=dev)
        self.motion_inr = NetworkWithInputEncoding(3, 2, encoding_config, mot_network_config, seed=seed, device=dev)
        self.masks = masks
        self.num_movements, self.x, self.num_lines = masks.shape
        self.device = dev
        # immoco.py:72-76 — computed with the same torch call on the same kind of device
        self.identy_grid = F.affine_grid(torch.eye(2, 3, device=dev).unsqueeze(0),
                                         torch.Size((1, 1, self.x, self.num_lines)), align_corners=True)
        self.input_grid = make_grids((self.num_movements, self.x, self.num_lines), device=dev)
        self.col_group = masks_to_col_group(masks) if self.num_movements > 0 else torch.zeros(
            self.num_lines, device=dev, dtype=torch.int32)

    def forward(self):
        H, W, nM = self.x, self.num_lines, self.num_movements
        o = self.image_inr(self.identy_grid.view(-1, 2)).float().view(H, W, 2)
        image_prior = torch.view_as_complex(o.contiguous())
        if nM > 0:
            grids = self.motion_inr(self.input_grid).float().tanh().view(nM, H, W, 2) \
                + self.identy_grid.view(1, H, W, 2)
            motion_images = _Warp.apply(image_prior, grids)
            kall = FFT(torch.cat([image_prior.unsqueeze(0), motion_images], dim=0))
        else:
            kall = FFT(image_prior.unsqueeze(0))
        kspace_out = _LineSelect.apply(kall, self.col_group)
        return kspace_out, image_prior


# ----------------------------------------------------------------------------------------------
# fused solver
# ----------------------------------------------------------------------------------------------
def mlp_mode(mlp_fp16) -> int:
    """immoco_solver_cfg.mlp_fp16 from the Python keyword: False / 0 -> exact fp32; True / 1 / "f16" -> tiny-cuda-nn's
    network precision (fp16 operands, fp32 accumulation, loss scale 128, fp16 activations between the kernels);
    "bf16x2" / 2 -> every matrix operand split into two bf16 terms (product error <= 2^-16.5), fp32 otherwise."""
    if isinstance(mlp_fp16, str):
        try:
            return {"f32": 0, "fp32": 0, "f16": 1, "fp16": 1, "f16mlp": 1, "bf16x2": 2}[mlp_fp16.lower()]
        except KeyError:
            raise L.ImmocoError(f"unknown MLP precision {mlp_fp16!r}") from None
    m = int(mlp_fp16)
    if m not in (0, 1, 2):
        raise L.ImmocoError(f"unknown MLP precision {mlp_fp16!r}")
    return m


class _SolverHandle:
    """RAII wrapper of immoco_solver_t; cached per (device, H, W, nM)."""

    def __init__(self, device, H, W, nM, use_graph=True, atomic_scatter=False, grad_parts=0, table_fp16=False,
                 batch_lanes=0, mlp_fp16=False, serial_chains=None, batch_pair=False):
        self.device, self.H, self.W, self.nM = device, H, W, nM
        self.image_grid = L.grid_cfg(2, encoding_config)
        self.motion_grid = L.grid_cfg(3, encoding_config)
        self.image_mlp = L.mlp_cfg(32, 2, network_config)
        self.motion_mlp = L.mlp_cfg(32, 2, mot_network_config)
        cfg = L.SolverCfg(H, W, nM, self.image_grid, self.motion_grid, self.image_mlp, self.motion_mlp,
                          1 if use_graph else 0, 1 if atomic_scatter else 0, int(grad_parts),
                          2 if serial_chains is None else (1 if serial_chains else 0), 1 if table_fp16 else 0, int(batch_lanes), mlp_mode(mlp_fp16),
                          1 if batch_pair else 0)
        self.handle = C.c_void_p()
        with torch.cuda.device(device):
            L.check(L.lib().immoco_solver_create(C.byref(cfg), C.byref(self.handle)), "solver_create")
        self.n_params_image = int(L.lib().immoco_solver_n_params(self.handle, 0))
        self.n_params_motion = int(L.lib().immoco_solver_n_params(self.handle, 1))
        # the reference's coordinate lattices (immoco.py:48-53,72-80), computed by torch on the device
        self.xs = torch.linspace(-1, 1, W, device=device)
        self.ys = torch.linspace(-1, 1, H, device=device)
        self.ms = torch.linspace(-1, 1, max(nM, 1), device=device)
        with torch.cuda.device(device):
            L.check(L.lib().immoco_solver_set_lattice(self.handle, L.ptr(self.xs), L.ptr(self.ys), L.ptr(self.ms),
                                                      L.stream_ptr()), "solver_set_lattice")

    @property
    def workspace_bytes(self):
        return int(L.lib().immoco_solver_workspace_bytes(self.handle))

    def close(self):
        if self.handle:
            L.lib().immoco_solver_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def init_params(self, seed_image=1337, seed_motion=1337):
        pi = torch.empty(self.n_params_image, device=self.device, dtype=torch.float32)
        pm = torch.empty(self.n_params_motion, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            st = L.stream_ptr()
            L.check(L.lib().immoco_init_params(C.byref(self.image_grid), C.byref(self.image_mlp),
                                               seed_image & 0xFFFFFFFF, L.ptr(pi), st), "init_params")
            L.check(L.lib().immoco_init_params(C.byref(self.motion_grid), C.byref(self.motion_mlp),
                                               seed_motion & 0xFFFFFFFF, L.ptr(pm), st), "init_params")
        return pi, pm

    def solve(self, kspace_norm, col_group, p_img, p_mot, a_img, a_mot, iters, lr, lambdas, step0=0,
              want_loss=False):
        dev = self.device
        out_img = torch.empty((self.H, self.W), device=dev, dtype=torch.complex64)
        out_k = torch.empty((self.H, self.W), device=dev, dtype=torch.complex64)
        loss = torch.empty(iters, device=dev, dtype=torch.float32) if want_loss else None
        lam = (C.c_float * iters)(*[float(v) for v in lambdas])
        with torch.cuda.device(dev):
            L.check(L.lib().immoco_solver_solve(
                self.handle, L.ptr(kspace_norm), L.ptr(col_group),
                L.ptr(p_img), L.ptr(p_mot), L.ptr(a_img), L.ptr(a_mot), iters, float(lr), lam, int(step0),
                L.ptr(out_img), L.ptr(out_k), L.ptr(loss), L.stream_ptr()), "solver_solve")
        return out_img, out_k, loss

    def solve_batch(self, kspace_norm, col_group, p_img, p_mot, a_img, a_mot, iters, lr, lambdas, step0=0,
                    want_loss=False):
        """B slices in one call: every tensor carries a leading batch dimension."""
        dev, B = self.device, int(kspace_norm.shape[0])
        out_img = torch.empty((B, self.H, self.W), device=dev, dtype=torch.complex64)
        out_k = torch.empty((B, self.H, self.W), device=dev, dtype=torch.complex64)
        loss = torch.empty((B, iters), device=dev, dtype=torch.float32) if want_loss else None
        lam = (C.c_float * iters)(*[float(v) for v in lambdas])
        with torch.cuda.device(dev):
            L.check(L.lib().immoco_solver_solve_batch(
                self.handle, B, L.ptr(kspace_norm), L.ptr(col_group),
                L.ptr(p_img), L.ptr(p_mot), L.ptr(a_img), L.ptr(a_mot), iters, float(lr), lam, int(step0),
                L.ptr(out_img), L.ptr(out_k), L.ptr(loss), L.stream_ptr()), "solver_solve_batch")
        return out_img, out_k, loss

    def forward(self, col_group, p_img, p_mot):
        out_img = torch.empty((self.H, self.W), device=self.device, dtype=torch.complex64)
        out_k = torch.empty((self.H, self.W), device=self.device, dtype=torch.complex64)
        with torch.cuda.device(self.device):
            L.check(L.lib().immoco_solver_forward(self.handle, L.ptr(col_group), L.ptr(p_img), L.ptr(p_mot),
                                                  L.ptr(out_img), L.ptr(out_k), L.stream_ptr()), "solver_forward")
        return out_k, out_img

    def profile(self, kspace_norm, col_group, p_img, p_mot, a_img, a_mot, reps=5, lr=1e-2, lambda_ge=1e-2):
        with torch.cuda.device(self.device):
            L.check(L.lib().immoco_solver_profile(
                self.handle, L.ptr(kspace_norm), L.ptr(col_group),
                L.ptr(p_img), L.ptr(p_mot), L.ptr(a_img), L.ptr(a_mot), reps, float(lr), float(lambda_ge),
                L.stream_ptr()), "solver_profile")
            torch.cuda.synchronize(self.device)
        names = (C.c_char_p * 64)()
        ms = (C.c_float * 64)()
        n = L.lib().immoco_solver_phase_times(self.handle, names, ms, 64)
        return [(names[i].decode(), float(ms[i])) for i in range(n)]

    def set_graph(self, on: bool):
        L.check(L.lib().immoco_solver_set_graph(self.handle, 1 if on else 0), "solver_set_graph")

    @property
    def dominant_kernel_ms(self):
        """Duration of the motion-grid encode backward in the last iteration of the last solve (in-graph events)."""
        return float(L.lib().immoco_solver_dominant_kernel_ms(self.handle))

    @property
    def graph_active(self):
        return bool(L.lib().immoco_solver_graph_active(self.handle))


_SOLVERS = {}


def get_solver(device, H, W, nM, use_graph=True, atomic_scatter=False, grad_parts=0, instance=0,
               table_fp16=False, batch_lanes=0, mlp_fp16=False, serial_chains=None, batch_pair=False) -> _SolverHandle:
    device = torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    key = (device.index, H, W, nM, bool(use_graph), bool(atomic_scatter), int(grad_parts), int(instance),
           bool(table_fp16), int(batch_lanes), mlp_mode(mlp_fp16), serial_chains, bool(batch_pair))
    s = _SOLVERS.get(key)
    if s is None:
        s = _SOLVERS[key] = _SolverHandle(device, H, W, nM, use_graph, atomic_scatter, grad_parts, table_fp16,
                                          batch_lanes, mlp_fp16, serial_chains, batch_pair)
    return s


def lambda_schedule(iters, lambda_ge, rule="immoco"):
    """GE weight used at iteration j.  rule="immoco": immoco.py:180-181 (halves on every j > iters//2
    that is NOT a multiple of iters//10; ZeroDivisionError for iters < 10, like the reference).
    rule="downstream": src/test/test_immoco_downstream.py:188-189 (j % 10 == 0 and j > 80)."""
    lam, out = float(lambda_ge), []
    for j in range(iters):
        out.append(lam)
        if rule == "immoco":
            if j % (iters // 10) and j > (iters // 2):
                lam *= 0.5
        elif rule == "downstream":
            if j % 10 == 0 and j > 80:
                lam *= 0.5
        else:
            raise ValueError(f"unknown lambda rule {rule!r}")
    return out


def imcoco_motion_correction(kspace_corr, masks, iters=200, learning_rate=1e-2, lambda_ge=1e-2, debug=False,
                             *, seed=1337, norm_scale=16000.0, lambda_rule="immoco", return_loss=False,
                             use_graph=True, atomic_scatter=False, grad_parts=0, instance=0, table_fp16=False,
                             mlp_fp16=False, serial_chains=None):
    """IM-MoCo per-slice solve (immoco.py:116-206).

    Args mirror the reference: ``kspace_corr`` [H, W] complex (any device), ``masks`` [nM, H, W]
    one-hot column masks on the GPU (their device decides where the solve runs, immoco.py:70).
    Returns ``(image_prior, kspace_foward_model)`` of the LAST forward pass, i.e. before the final
    Adam step (immoco.py:203-206).  Keyword-only extras expose the downstream script's variants
    (``norm_scale=8000``, ``lambda_rule="downstream"``; test_immoco_downstream.py:150-152,188-189).
    The call returns as soon as the work is queued on the solver's own stream (the outputs are
    ordered after it on the caller's stream); ``instance`` selects one of several solver handles of
    the same shape, so that independent slices can be in flight concurrently on one GPU;
    ``table_fp16=True`` gathers the hash-grid features from fp16 shadows of the fp32 master tables
    (tiny-cuda-nn's own precision; BASELINE config 5); ``mlp_fp16=True`` runs both MLPs with fp16 operands and
    fp32 accumulation (tcnn's network precision, loss scale 128) instead of exact fp32.
    """
    L.require_gpu(masks, what="imcoco_motion_correction(masks)")
    dev = masks.device
    nM, H, W = masks.shape
    lambdas = lambda_schedule(iters, lambda_ge, lambda_rule)   # raises ZeroDivisionError like the reference
    solver = get_solver(dev, H, W, nM, use_graph, atomic_scatter, grad_parts, instance, table_fp16,
                        mlp_fp16=mlp_fp16, serial_chains=serial_chains)
    k = kspace_corr.to(dev).to(torch.complex64).contiguous()
    if k.shape != (H, W):
        raise L.ImmocoError(f"kspace_corr shape {tuple(k.shape)} does not match masks {(H, W)}")
    kin = torch.empty_like(k)
    scale = torch.empty(1, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        L.check(L.lib().immoco_normalize_kspace(L.ptr(k), H * W, float(norm_scale), L.ptr(kin), L.ptr(scale),
                                                L.stream_ptr()), "normalize_kspace")
    col_group = masks_to_col_group(masks) if nM > 0 else torch.zeros(W, device=dev, dtype=torch.int32)
    p_img, p_mot = solver.init_params(seed, seed)
    a_img = torch.zeros(2 * solver.n_params_image, device=dev, dtype=torch.float32)
    a_mot = torch.zeros(2 * solver.n_params_motion, device=dev, dtype=torch.float32)
    image_prior, kfm, loss = solver.solve(kin, col_group, p_img, p_mot, a_img, a_mot, iters, learning_rate,
                                          lambdas, want_loss=bool(debug or return_loss))
    if debug:
        print(f"Scale: {scale.item():.4f}")
        lh = loss.cpu()
        for j in range(0, iters, 20):
            print(f"iter: {j}, DC_Loss: {lh[j]:.4f}")
    if return_loss:
        return image_prior, kfm, loss
    return image_prior, kfm


def imcoco_motion_correction_batch(kspaces, masks_list, iters=200, learning_rate=1e-2, lambda_ge=1e-2, *, seed=1337,
                                   norm_scale=16000.0, lambda_rule="immoco", return_loss=False, use_graph=True,
                                   table_fp16=False, lanes=1, mlp_fp16=False, pair=False, serial_chains=None):
    """``imcoco_motion_correction`` for a batch (BASELINE config 3: B slices resident on one GPU):
    ``kspaces [B, H, W] c64`` and one ``masks [nM_i, H, W]`` per slice.  Slices with the same number of
    movement groups share one ``immoco_solver_solve_batch`` call (parameters, Adam state and outputs live in
    ``[B_g, ...]`` tensors; 305 MB of fp32 state per slice); ``lanes`` > 1 keeps that many slices in flight side by
    side (``immoco_solver_cfg.batch_lanes``; slower than slice after slice on MI355X); ``pair=True``
    solves two slices at a time inside one graph with their hash-grid gathers serialised by events
    (``immoco_solver_cfg.batch_pair``).  Returns ``(image_prior [B, H, W],
    kspace_foward_model [B, H, W])`` (+ ``loss [B, iters]`` with ``return_loss``), slice i initialised like a
    single call with the same ``seed``."""
    if kspaces.dim() != 3 or len(masks_list) != kspaces.shape[0]:
        raise L.ImmocoError("kspaces must be [B, H, W] with one masks tensor per slice")
    B, H, W = kspaces.shape
    L.require_gpu(kspaces, *masks_list, what="imcoco_motion_correction_batch")
    dev = kspaces.device
    lambdas = lambda_schedule(iters, lambda_ge, lambda_rule)
    images = torch.empty((B, H, W), device=dev, dtype=torch.complex64)
    kfms = torch.empty_like(images)
    losses = torch.empty((B, iters), device=dev, dtype=torch.float32) if return_loss else None
    by_nm = {}
    for i, m in enumerate(masks_list):
        if m.dim() != 3 or tuple(m.shape[1:]) != (H, W):
            raise L.ImmocoError(f"masks[{i}] shape {tuple(m.shape)} does not match kspaces {(H, W)}")
        by_nm.setdefault(int(m.shape[0]), []).append(i)
    for nM, idx in sorted(by_nm.items()):
        solver = get_solver(dev, H, W, nM, use_graph, False, 0, 0, table_fp16, lanes, mlp_fp16=mlp_fp16,
                            serial_chains=serial_chains, batch_pair=pair)
        Bg = len(idx)
        k = kspaces[idx].to(torch.complex64).contiguous()
        kin = torch.empty_like(k)
        scale = torch.empty(1, device=dev, dtype=torch.float32)
        cgs = torch.zeros((Bg, W), device=dev, dtype=torch.int32)
        p_img = torch.empty((Bg, solver.n_params_image), device=dev, dtype=torch.float32)
        p_mot = torch.empty((Bg, max(solver.n_params_motion, 1)), device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            for j, i in enumerate(idx):
                L.check(L.lib().immoco_normalize_kspace(L.ptr(k[j]), H * W, float(norm_scale), L.ptr(kin[j]),
                                                        L.ptr(scale), L.stream_ptr()), "normalize_kspace")
                if nM > 0:
                    cgs[j] = masks_to_col_group(masks_list[i].to(dev))
        pi0, pm0 = solver.init_params(seed, seed)
        p_img[:] = pi0
        p_mot = pm0.expand(Bg, -1).contiguous() if nM > 0 else p_mot
        a_img = torch.zeros((Bg, 2 * solver.n_params_image), device=dev, dtype=torch.float32)
        a_mot = torch.zeros((Bg, 2 * solver.n_params_motion), device=dev, dtype=torch.float32)
        im, kf, ls = solver.solve_batch(kin, cgs, p_img, p_mot, a_img, a_mot, iters, learning_rate, lambdas,
                                        want_loss=return_loss)
        ii = torch.as_tensor(idx, device=dev)
        images[ii], kfms[ii] = im, kf
        if return_loss:
            losses[ii] = ls
    return (images, kfms, losses) if return_loss else (images, kfms)
