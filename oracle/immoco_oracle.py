"""CPU ORACLE for the IM-MoCo per-slice inner optimisation loop.

THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT.  It is a plain torch-CPU / numpy
restatement of the reference algorithm.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it; the product path (``miccai24_immoco_amd``) never does and fails
loudly when the HIP library is missing.

Parity status
-------------
* Everything that lives under ``/root/reference`` (grids, warp, centred FFT,
  line select, losses, Adam loop, lambda schedule, return values, mask builder,
  motion simulator, PSNR) is PINNED: ``tools/gen_golden.py`` imports the
  reference's own Python in the build container and the resulting vectors are
  committed under ``tests/golden/`` (``tests/test_oracle_golden.py``).
* The INR arithmetic (multiresolution hash grid + bias-free MLP) lives in the
  un-vendored, un-pinned third-party dependency **tiny-cuda-nn** (installed from
  ``git+https://github.com/NVlabs/tiny-cuda-nn`` HEAD, reference README.md:56-60;
  call sites src/models/immoco.py:1,60-65,85,93).  Its source is not in
  ``/root/reference`` and it cannot run here (CUDA only), so that part restates
  the published algorithm (Mueller et al. 2022, "Instant neural graphics
  primitives", and SURVEY.md Appendix A) in fp32 and is **parity unpinned**
  against tiny-cuda-nn itself; it is pinned only by known-answer tests
  (tests/test_oracle_hashgrid.py).

Every function cites the reference file:line it follows (paths relative to
``/root/reference``).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# configs: src/models/immoco.py:11-37 (verbatim keys; "fine_resolution" is not a
# tiny-cuda-nn key and is ignored upstream)
# --------------------------------------------------------------------------
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

PRIMES = (1, 2654435761, 805459861)  # tiny-cuda-nn coherent_prime_hash, first 3
U32 = 0xFFFFFFFF


# --------------------------------------------------------------------------
# hash-grid geometry (tiny-cuda-nn grid.h semantics)
# --------------------------------------------------------------------------
@dataclass
class GridGeometry:
    dims: int
    n_levels: int = 16
    n_features: int = 2
    log2_hashmap_size: int = 19
    base_resolution: int = 16
    per_level_scale: float = 2.0
    scales: List[float] = field(default_factory=list)
    resolutions: List[int] = field(default_factory=list)
    sizes: List[int] = field(default_factory=list)      # entries per level
    offsets: List[int] = field(default_factory=list)    # entry offset per level (+ total)
    hashed: List[bool] = field(default_factory=list)

    def __post_init__(self):
        log2_pls = np.float32(np.log2(np.float32(self.per_level_scale)))
        off = 0
        for l in range(self.n_levels):
            # grid_scale(): exp2f(level*log2_per_level_scale)*base_resolution - 1.0f
            scale = np.float32(np.exp2(np.float32(l) * log2_pls)) * np.float32(
                self.base_resolution) - np.float32(1.0)
            res = int(np.ceil(scale)) + 1           # grid_resolution()
            dense = res ** self.dims
            n = min(dense, 0x7FFFFFFF)
            n = (n + 7) // 8 * 8                      # next_multiple(.,8)
            n = min(n, 1 << self.log2_hashmap_size)  # GridType::Hash
            # grid_index(): stride after the loop vs hashmap_size decides hashing.  Upstream walks
            # `uint32_t stride; for (dim < N_DIMS && stride <= hashmap_size) stride *= resolution;`
            # i.e. the product WRAPS mod 2^32.  With base 16 / per_level_scale 2 the resolutions are
            # powers of two, so from res = 2^16 (level 12) on res*res wraps to exactly 0,
            # `hashmap_size < stride` is false and the level uses the dense wrapped index
            # (c0 + c1*res [+ c2*0]) % hashmap_size instead of the prime hash (SURVEY A.3 "(uint32 wrap)").
            stride = 1
            for _ in range(self.dims):
                if stride > n:
                    break
                stride = (stride * res) & U32
            self.scales.append(float(scale))
            self.resolutions.append(res)
            self.sizes.append(n)
            self.offsets.append(off)
            self.hashed.append(n < stride)
            off += n
        self.offsets.append(off)

    @property
    def n_entries(self) -> int:
        return self.offsets[-1]

    @property
    def n_table_params(self) -> int:
        return self.n_entries * self.n_features

    @property
    def enc_width(self) -> int:
        return self.n_levels * self.n_features


def geometry_from_config(dims: int, enc_cfg: dict) -> GridGeometry:
    assert enc_cfg.get("otype", "Grid").lower() in ("grid", "hashgrid")
    assert enc_cfg.get("type", "Hash").lower() == "hash"
    assert enc_cfg.get("interpolation", "Linear").lower() == "linear"
    return GridGeometry(
        dims=dims,
        n_levels=int(enc_cfg.get("n_levels", 16)),
        n_features=int(enc_cfg.get("n_features_per_level", 2)),
        log2_hashmap_size=int(enc_cfg.get("log2_hashmap_size", 19)),
        base_resolution=int(enc_cfg.get("base_resolution", 16)),
        per_level_scale=float(enc_cfg.get("per_level_scale", 2.0)),
    )


def _fmaf(a: np.ndarray, b: float, c: float) -> np.ndarray:
    """float32 fmaf(b, a, c) emulated through float64 (products of two f32 are
    exact in f64; the single add is exact for the magnitudes used here)."""
    return (a.astype(np.float64) * np.float64(np.float32(b)) + np.float64(c)).astype(np.float32)


def grid_cells(coords: np.ndarray, geo: GridGeometry, level: int):
    """pos_fract(): pos = fmaf(scale, x, 0.5); cell = floor(pos); w = pos - cell;
    cell_u = (uint32)(int32)cell  (SURVEY Appendix A.3).  coords [N,D] float32.
    Returns cell_u [N,D] uint64 (values < 2^32) and frac [N,D] float32."""
    pos = _fmaf(coords.astype(np.float32), geo.scales[level], 0.5)
    fl = np.floor(pos)
    frac = (pos - fl).astype(np.float32)
    cell = fl.astype(np.int64) & U32  # two's complement wrap of negative cells
    return cell.astype(np.uint64), frac


def grid_index(cell_u: np.ndarray, geo: GridGeometry, level: int) -> np.ndarray:
    """grid_index(): dense stride walk with uint32 wrap, or coherent-prime hash,
    then `% hashmap_size`.  cell_u [...,D] uint64 (<2^32) -> index int64."""
    n = geo.sizes[level]
    res = geo.resolutions[level]
    if geo.hashed[level]:
        idx = np.zeros(cell_u.shape[:-1], dtype=np.uint64)
        for d in range(geo.dims):
            idx ^= (cell_u[..., d] * np.uint64(PRIMES[d])) & np.uint64(U32)
    else:
        idx = np.zeros(cell_u.shape[:-1], dtype=np.uint64)
        stride = 1                      # uint32 like upstream: the product wraps
        for d in range(geo.dims):
            if stride > n:
                break
            idx = (idx + cell_u[..., d] * np.uint64(stride)) & np.uint64(U32)
            stride = (stride * res) & U32
    return (idx % np.uint64(n)).astype(np.int64)


def grid_corners(coords: np.ndarray, geo: GridGeometry, level: int):
    """All 2^D corners of every point at one level.
    Returns idx [N, 2^D] int64 (entry index inside the level) and w [N, 2^D] f32.
    Corner bit d set => cell+1 with weight frac[d], else cell with 1-frac[d];
    weights multiplied in dim order in fp32 (grid.h kernel_grid)."""
    cell, frac = grid_cells(coords, geo, level)
    N, D = coords.shape
    idxs, ws = [], []
    for corner in range(1 << D):
        w = np.ones(N, dtype=np.float32)
        c = cell.copy()
        for d in range(D):
            if corner & (1 << d):
                w = (w * frac[:, d]).astype(np.float32)
                c[:, d] = (c[:, d] + np.uint64(1)) & np.uint64(U32)
            else:
                w = (w * (np.float32(1.0) - frac[:, d])).astype(np.float32)
        idxs.append(grid_index(c, geo, level))
        ws.append(w)
    return np.stack(idxs, 1), np.stack(ws, 1)


_HG_LIB = None


def _hg_lib():
    """ctypes handle of oracle/_build/libhashgrid_oracle.so (oracle/hashgrid_oracle.c), built on demand with
    gcc (oracle/Makefile).  TEST INFRASTRUCTURE like the rest of this file."""
    global _HG_LIB
    if _HG_LIB is None:
        import ctypes
        import os
        import subprocess
        here = os.path.dirname(os.path.abspath(__file__))
        so = os.path.join(here, "_build", "libhashgrid_oracle.so")
        src = os.path.join(here, "hashgrid_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.run(["make", "-C", here], check=True, stdout=subprocess.DEVNULL)
        h = ctypes.CDLL(so)
        P, I64, I32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
        h.hg_encode_fwd.argtypes = [P, P, P, I64, I32, I32, P]
        h.hg_encode_fwd.restype = None
        h.hg_encode_bwd.argtypes = [P, P, P, I64, I32, I32, P, I32]
        h.hg_encode_bwd.restype = None
        _HG_LIB = h
    return _HG_LIB


class _EncodeC(torch.autograd.Function):
    """enc = sum_c table[idx_c] * w_c and its transpose through oracle/hashgrid_oracle.c (same arithmetic as
    the torch expression in HashGridPlan.encode_torch: fp32, multiply then add in corner order)."""

    @staticmethod
    def forward(ctx, table, plan):
        t = table.detach().contiguous()
        enc = torch.empty((plan.n_points, plan.geo.n_levels * plan.geo.n_features), dtype=torch.float32)
        L, C = plan.idx_lm.shape[0], plan.idx_lm.shape[2]
        _hg_lib().hg_encode_fwd(t.data_ptr(), plan.idx_lm.data_ptr(), plan.w_lm.data_ptr(), plan.n_points, L, C,
                                enc.data_ptr())
        ctx.plan = plan
        ctx.shape = tuple(table.shape)
        return enc

    @staticmethod
    def backward(ctx, denc):
        plan = ctx.plan
        d = denc.contiguous()
        dt = torch.zeros(ctx.shape, dtype=torch.float32)
        L, C = plan.idx_lm.shape[0], plan.idx_lm.shape[2]
        _hg_lib().hg_encode_bwd(d.data_ptr(), plan.idx_lm.data_ptr(), plan.w_lm.data_ptr(), plan.n_points, L, C,
                                dt.data_ptr(), int(plan.bwd_order))
        return dt, None


class HashGridPlan:
    """Pre-computed (entry index, weight) for a FIXED set of coordinates.
    The reference always queries the same lattice (immoco.py:72-80), so the
    oracle computes the integer part once with numpy; the (linear) interpolation
    enc = sum_c w_c * table[idx_c] is evaluated either by a torch expression with
    torch autograd (`encode_torch`, the original restatement) or by the plain-C
    loop of oracle/hashgrid_oracle.c (`encode_c`: the same arithmetic, 3x faster per
    oracle iteration; checked against the torch path in tests/test_oracle_hashgrid.py).
    `bwd_order` selects the (deterministic) fp32 summation order of the C backward."""

    def __init__(self, coords: torch.Tensor, geo: GridGeometry, bwd_order: int = 0, device=None):
        c = coords.detach().cpu().to(torch.float32).numpy()
        self.geo = geo
        self.n_points = c.shape[0]
        self.bwd_order = bwd_order
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        self._dev_idx = None   # per-corner [N*L] int32 entry indices / [N, L, 1] weights on `device` (encode_device)
        self._dev_w = None
        idx_all, w_all = [], []
        for l in range(geo.n_levels):
            idx, w = grid_corners(c, geo, l)
            idx_all.append((idx + geo.offsets[l]).astype(np.int32))
            w_all.append(w)
        self.idx_lm = torch.from_numpy(np.stack(idx_all, 0))  # [L, N, 2^D] int32 (absolute entry index)
        self.w_lm = torch.from_numpy(np.stack(w_all, 0))      # [L, N, 2^D] f32

    @property
    def idx(self) -> torch.Tensor:
        """[N, L, 2^D] int64 view of the plan (point-major, as the torch expression indexes it)."""
        return self.idx_lm.permute(1, 0, 2).long()

    @property
    def w(self) -> torch.Tensor:
        return self.w_lm.permute(1, 0, 2)

    def encode_torch(self, table: torch.Tensor) -> torch.Tensor:
        """table [n_entries, F] -> enc [N, L*F] (level-major, feature-minor).
        Corner contributions accumulated in corner order, fp32."""
        g = self.geo
        idx, w = self.idx, self.w
        enc = None
        for corner in range(idx.shape[2]):
            t = table[idx[:, :, corner]] * w[:, :, corner, None]
            enc = t if enc is None else enc + t
        return enc.reshape(self.n_points, g.n_levels * g.n_features)

    def encode_c(self, table: torch.Tensor) -> torch.Tensor:
        return _EncodeC.apply(table, self)

    def encode_device(self, table: torch.Tensor) -> torch.Tensor:
        """The same expression as `encode_torch` evaluated by ATen on `table.device` (the DEVICE ORACLE, a
        sampler for the statistical parity tests: it shares no kernel with libimmoco_hip.so).  One
        `index_select` per corner: its autograd backward is `index_add_`, which ATen implements on the GPU
        with fp32 atomicAdd in nondeterministic order - the natural model of tiny-cuda-nn's atomic scatter
        (SURVEY A.5), where the CPU paths use one fixed (or per-step re-drawn) summation order."""
        g = self.geo
        if self._dev_idx is None or self._dev_idx[0].device != table.device:
            idx = self.idx_lm.permute(1, 0, 2)                       # [N, L, C]
            w = self.w_lm.permute(1, 0, 2)
            C = idx.shape[2]
            self._dev_idx = [idx[:, :, c].reshape(-1).contiguous().to(table.device) for c in range(C)]
            self._dev_w = [w[:, :, c].contiguous().to(table.device).unsqueeze(-1) for c in range(C)]
        enc = None
        for ic, wc in zip(self._dev_idx, self._dev_w):
            t = table.index_select(0, ic).view(self.n_points, g.n_levels, g.n_features) * wc
            enc = t if enc is None else enc + t
        return enc.reshape(self.n_points, g.n_levels * g.n_features)

    def encode(self, table: torch.Tensor, backend: str = "c") -> torch.Tensor:
        assert self.geo.n_features == 2
        if table.device.type != "cpu":
            return self.encode_device(table)
        if backend == "c" and table.dtype == torch.float32:
            return self.encode_c(table)
        return self.encode_torch(table)


# --------------------------------------------------------------------------
# parameter initialisation shared (bit-exact) by the oracle and the HIP path.
# tiny-cuda-nn: encoding U(-1e-4,1e-4); MLP Xavier-uniform with the PADDED
# output width.  tiny-cuda-nn's own RNG stream is not
# reproducible here, so both sides use the same counter-based generator
# (PCG output hash of seed/index); distribution parity only.
# --------------------------------------------------------------------------
def pcg_hash_u32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64) & np.uint64(U32)
    state = (x * np.uint64(747796405) + np.uint64(2891336453)) & np.uint64(U32)
    shift = (state >> np.uint64(28)) + np.uint64(4)
    word = (((state >> shift) ^ state) * np.uint64(277803737)) & np.uint64(U32)
    return ((word >> np.uint64(22)) ^ word) & np.uint64(U32)


def uniform_init(n: int, seed: int, stream: int, lo: float, hi: float, start: int = 0) -> np.ndarray:
    """value[i] = lo + u*(hi-lo), u = (pcg(pcg(seed^stream*0x9E3779B9) + i) >> 8) * 2^-24."""
    key = pcg_hash_u32(np.array([(seed ^ (stream * 0x9E3779B9)) & U32], dtype=np.uint64))[0]
    i = (np.arange(start, start + n, dtype=np.uint64) + key) & np.uint64(U32)
    u = (pcg_hash_u32(i) >> np.uint64(8)).astype(np.float32) * np.float32(2.0 ** -24)
    return (np.float32(lo) + u * np.float32(np.float32(hi) - np.float32(lo))).astype(np.float32)


@dataclass
class MLPSpec:
    n_in: int
    n_hidden: int
    n_out: int
    n_out_padded: int
    activation: str  # "relu" | "tanh"

    @property
    def n_w1(self):
        return self.n_hidden * self.n_in

    @property
    def n_w2(self):
        return self.n_out_padded * self.n_hidden

    @property
    def n_params(self):
        return self.n_w1 + self.n_w2


def mlp_spec_from_config(n_in: int, n_out: int, net_cfg: dict) -> MLPSpec:
    otype = net_cfg["otype"].lower()
    assert int(net_cfg.get("n_hidden_layers", 1)) == 1, "reference uses one hidden layer"
    assert net_cfg.get("output_activation", "None").lower() == "none"
    pad = 16 if otype == "fullyfusedmlp" else 8
    act = net_cfg["activation"].lower()
    assert act in ("relu", "tanh")
    return MLPSpec(n_in, int(net_cfg["n_neurons"]), n_out, (n_out + pad - 1) // pad * pad, act)


def init_inr_params(geo: GridGeometry, mlp: MLPSpec, seed: int) -> np.ndarray:
    """Flat fp32 params in tiny-cuda-nn order: [W1 (hidden x in) | W2 (padded_out x hidden) | table]."""
    b1 = math.sqrt(6.0 / (mlp.n_in + mlp.n_hidden))
    b2 = math.sqrt(6.0 / (mlp.n_hidden + mlp.n_out_padded))
    w1 = uniform_init(mlp.n_w1, seed, 1, -b1, b1)
    w2 = uniform_init(mlp.n_w2, seed, 2, -b2, b2)
    tab = uniform_init(geo.n_table_params, seed, 3, -1e-4, 1e-4)
    return np.concatenate([w1, w2, tab])


class _MLPHalf(torch.autograd.Function):
    """The bias-free MLP in tiny-cuda-nn's OWN operand precision (reference call sites
    src/models/immoco.py:11-25,60-65: FullyFusedMLP / CutlassMLP are instantiated with `__half` network
    precision): every matrix-product operand is rounded to fp16 - the encoding, W1, the hidden activations, W2,
    and in the backward dL/dout * loss_scale (tcnn's torch binding: loss_scale = 128) and dL/dpre - every
    product is accumulated in fp32 (tensor cores / MFMA), activations are evaluated in fp32.  Outputs and the
    weight gradients leave in fp32 (tcnn narrows them to fp16 as well; this mode is never narrower than tcnn).
    `denc_fp16`: dL/denc * loss_scale is rounded to fp16 before it is unscaled - what tcnn hands to its grid
    backward, and what the solver stores between its MLP and encode backward kernels (the op-level C-ABI keeps
    fp32 there).  The derivative of the activation is taken from the STORED fp16 activation, like tcnn's backward."""

    @staticmethod
    def forward(ctx, enc, w1, w2, act, loss_scale, denc_fp16=False):
        e16, w1h, w2h = enc.half().float(), w1.half().float(), w2.half().float()
        pre = e16 @ w1h.t()
        h16 = (torch.relu(pre) if act == "relu" else torch.tanh(pre)).half().float()
        ctx.save_for_backward(e16, w1h, w2h, h16)
        ctx.act, ctx.S, ctx.denc_fp16 = act, float(loss_scale), bool(denc_fp16)
        return h16 @ w2h.t()

    @staticmethod
    def backward(ctx, dout):
        e16, w1h, w2h, h16 = ctx.saved_tensors
        S = ctx.S
        d16 = (dout * S).half().float()
        dact = (h16 > 0).float() if ctx.act == "relu" else 1.0 - h16 * h16
        dp16 = ((d16 @ w2h) * dact).half().float()
        denc = dp16 @ w1h
        if ctx.denc_fp16:
            denc = denc.half().float()
        return denc / S, (dp16.t() @ e16) / S, (d16.t() @ h16) / S, None, None, None


def tanh_alt(x: torch.Tensor) -> torch.Tensor:
    """An equally valid fp32 tanh: 1 - 2 / (exp(2|x|) + 1) with x - x^3/3 below 0.04 (the formula of the HIP kernels,
    csrc/mlp_mfma.hip:tanh_fast, evaluated with torch's exp) - differs from torch.tanh by ~1e-7.  Used by the sensitivity
    experiment of DESIGN.md 2.4 only (OracleINR(tanh="alt")): does an O(1e-7) detail of the ORACLE move its own PSNR level?"""
    ax = x.abs()
    big = 1.0 - 2.0 / (torch.exp(2.0 * ax) + 1.0)
    small = ax * (1.0 - ax * ax * 0.33333334)
    return torch.where(ax < 0.04, small, big) * torch.sign(x)


class OracleINR(torch.nn.Module):
    """Stand-in for tinycudann.NetworkWithInputEncoding(n_in, n_out, enc_cfg, net_cfg)
    (call sites immoco.py:60-65,85,93) in fp32 on the CPU.  One flat `params`
    Parameter like the tcnn torch binding.  The coordinate plan is cached on the
    first forward (the reference always passes the same grid)."""

    def __init__(self, n_input_dims, n_output_dims, encoding_config, network_config, seed=1337, table_fp16=False,
                 backend="c", bwd_order=0, mlp_fp16=False, loss_scale=128.0, denc_fp16=True, device=None, tanh="torch",
                 mlp_f64=False, mlp_splitk=0):
        super().__init__()
        self.tanh = tanh_alt if tanh == "alt" else torch.tanh
        # mlp_f64: the three matrix products of the MLP and their autograd transposes accumulate in float64 (operands are the
        # fp32 values, results are rounded to fp32 once): the MLP "without GEMM rounding noise" of the sensitivity
        # experiment of DESIGN.md 2.4 (the HIP kernels' fused-multiply-add chains are closer to this than to a blocked fp32 GEMM)
        self.mlp_f64 = mlp_f64
        # mlp_splitk = c > 1: every matrix product of the MLP (and, through autograd, of its backward) is evaluated as the sum
        # of c products over interleaved slices k = j mod c of its inner dimension - another equally valid fp32 evaluation
        # order of the same sums (what a different GEMM tiling does); sensitivity experiment of DESIGN.md 2.4
        self.mlp_splitk = int(mlp_splitk)
        # device != cpu: the DEVICE ORACLE (ATen kernels only; HashGridPlan.encode_device) - the fast sampler
        # of the statistical parity fixtures (tools/device_oracle_sampler.py); validated against the CPU oracle
        # by tests/test_gpu_ops.py::test_device_oracle_vs_cpu_oracle_teacher_forced
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        self.table_fp16 = table_fp16   # gather from an fp16 copy of the table (fp32 master, straight-through)
        self.mlp_fp16 = mlp_fp16       # fp16 MLP operands, fp32 accumulation (_MLPHalf): tcnn's network precision
        self.loss_scale = loss_scale
        self.denc_fp16 = denc_fp16     # with mlp_fp16: dL/denc * loss_scale rounded to fp16 (the solver's layout)
        self.backend = backend         # "c": oracle/hashgrid_oracle.c; "torch": the torch expression + autograd
        self.bwd_order = bwd_order     # summation order of the C backward (HashGridPlan)
        self.geo = geometry_from_config(n_input_dims, encoding_config)
        self.mlp = mlp_spec_from_config(self.geo.enc_width, n_output_dims, network_config)
        self.n_output_dims = n_output_dims
        self.seed = seed
        p = init_inr_params(self.geo, self.mlp, seed)
        self.params = torch.nn.Parameter(torch.from_numpy(p).to(self.device))
        self._plan: Optional[HashGridPlan] = None
        self._plan_key = None
        self._row_perm = None          # redraw(): row permutation of the MLP's batch (dW summation order)

    def redraw(self, rng: np.random.Generator):
        """Draw NEW fp32 summation orders for the next step: the block order of the hash-grid backward
        (oracle/hashgrid_oracle.c) and the row order of the MLP batch (the order in which dW1 / dW2 sum over the
        points).  Every order is an equally valid evaluation of the same algorithm - this is what the
        nondeterministic atomics of tiny-cuda-nn's backward do on EVERY step, where a fixed
        `bwd_order` keeps one order for a whole trajectory."""
        self.bwd_order = int(rng.integers(2, 4096))
        if self._plan is not None:
            self._plan.bwd_order = self.bwd_order
            n, blk = self._plan.n_points, 1024
            if n % blk == 0:
                bp = torch.from_numpy(rng.permutation(n // blk))
                self._row_perm = (bp[:, None] * blk + torch.arange(blk)[None, :]).reshape(-1)
            else:
                self._row_perm = torch.from_numpy(rng.permutation(n))
            self._row_perm = self._row_perm.to(self.device)

    def split(self, params=None):
        p = self.params if params is None else params
        m = self.mlp
        w1 = p[: m.n_w1].view(m.n_hidden, m.n_in)
        w2 = p[m.n_w1: m.n_params].view(m.n_out_padded, m.n_hidden)
        tab = p[m.n_params:].view(self.geo.n_entries, self.geo.n_features)
        return w1, w2, tab

    def plan_for(self, x: torch.Tensor) -> HashGridPlan:
        key = (x.data_ptr(), tuple(x.shape))
        if self._plan is None or self._plan_key != key:
            self._plan = HashGridPlan(x, self.geo, self.bwd_order, device=self.device)
            self._plan_key = key
        return self._plan

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        w1, w2, tab = self.split()
        if self.table_fp16:
            tab = tab + (tab.half().float() - tab).detach()
        enc = self.plan_for(x).encode(tab, self.backend)
        perm = self._row_perm
        if perm is not None:
            enc = enc[perm]
        if self.mlp_fp16:
            out = _MLPHalf.apply(enc, w1, w2, self.mlp.activation, self.loss_scale, self.denc_fp16)
        else:
            if self.mlp_f64:
                pre = (enc.double() @ w1.double().t()).float()
                h = torch.relu(pre) if self.mlp.activation == "relu" else self.tanh(pre)
                out = (h.double() @ w2.double().t()).float()
            elif self.mlp_splitk > 1:
                c = self.mlp_splitk
                pre = sum(enc[:, j::c] @ w1[:, j::c].t() for j in range(c))
                h = torch.relu(pre) if self.mlp.activation == "relu" else self.tanh(pre)
                out = sum(h[:, j::c] @ w2[:, j::c].t() for j in range(c))
            else:
                pre = enc @ w1.t()
                h = torch.relu(pre) if self.mlp.activation == "relu" else self.tanh(pre)
                out = h @ w2.t()
        if perm is not None:
            out = torch.empty_like(out).index_copy(0, perm, out)
        return out[:, : self.n_output_dims]


# --------------------------------------------------------------------------
# operators under /root/reference (pinned by tests/golden)
# --------------------------------------------------------------------------
def FFT(x):
    """src/utils/data_utils.py:29-30 — centred, unnormalised 2-D FFT."""
    return torch.fft.fftshift(
        torch.fft.fftn(torch.fft.ifftshift(x, dim=(-2, -1)), dim=(-2, -1)), dim=(-2, -1))


def IFFT(x):
    """src/utils/data_utils.py:33-34 — centred, 1/(HW)-normalised inverse."""
    return torch.fft.ifftshift(
        torch.fft.ifftn(torch.fft.fftshift(x, dim=(-2, -1)), dim=(-2, -1)), dim=(-2, -1))


def gradient_entropy_loss(x: torch.Tensor) -> torch.Tensor:
    """src/utils/losses.py:20-40."""
    dx = (x[:, :-1] - x[:, 1:]).abs()
    dy = (x[:-1, :] - x[1:, :]).abs()
    dx = F.pad(dx, (0, 1, 0, 0), mode="constant", value=0)
    dy = F.pad(dy, (0, 0, 0, 1), mode="constant", value=0)
    g = dx + dy
    return -torch.sum(g * torch.log(g + 1e-24))


def make_grids(sizes, device="cpu"):
    """src/models/immoco.py:48-53."""
    lin = [torch.linspace(-1, 1, s, device=device) for s in sizes]
    mesh = torch.meshgrid(*lin, indexing="ij")
    return torch.stack(mesh, dim=-1).view(-1, len(sizes))


def extract_movement_groups(lines: torch.Tensor, make_list: bool = False) -> torch.Tensor:
    """src/utils/motion_utils.py:56-109 — run-length labelling of corrupted
    phase-encode lines.  Vectorised restatement: a line belongs to group
    1 + (number of True->False falling edges strictly before it)."""
    v = np.asarray(lines.detach().cpu()).astype(bool)
    n = v.shape[0]
    labels = np.zeros(n, dtype=np.int64)
    count = 1
    for i in range(n):
        if v[i]:
            labels[i] = count
            if i != n - 1 and not v[i + 1]:
                count += 1
    groups = torch.from_numpy(np.broadcast_to(labels[None, :], (n, n)).copy())
    if not make_list:
        return groups
    # torch.unique(groups).nonzero().squeeze().numel(): number of non-zero labels
    # (NB when no line is corrupted unique=[0] -> counts 0; labels are contiguous)
    uniq = np.unique(labels)
    counts = int(np.count_nonzero(uniq))
    out = torch.zeros((counts, n, n), dtype=torch.long)
    for i in range(counts):
        out[i][groups == i + 1] = 1
    return out


def col_group_from_masks(masks: torch.Tensor) -> torch.Tensor:
    """Compact form of the line-select masks: g[c] in {0..nM}, 0 = unmasked.
    Valid because extract_movement_groups masks are constant down each column
    (motion_utils.py:74-91)."""
    nM = masks.shape[0]
    w = torch.arange(1, nM + 1, dtype=torch.long).view(nM, 1)
    return (masks[:, 0, :].long() * w).sum(0).to(torch.int32)


def identity_grid(H: int, W: int) -> torch.Tensor:
    """src/models/immoco.py:72-76 — affine_grid(eye, align_corners=True): [1,H,W,2] (x,y)."""
    return F.affine_grid(torch.eye(2, 3).unsqueeze(0), torch.Size((1, 1, H, W)), align_corners=True)


class OracleIMMoCo(torch.nn.Module):
    """src/models/immoco.py:56-113 with the INRs injected."""

    def __init__(self, masks, image_inr=None, motion_inr=None, seed=1337, device=None):
        super().__init__()
        # both INRs start from the SAME seed: tiny-cuda-nn's module default is seed=1337 for every module
        # (immoco.py:60-65 pass none), and every fixture and the HIP path (immoco_init_params) use 1337 for both
        self.image_inr = image_inr or OracleINR(2, 2, encoding_config, network_config, seed=seed, device=device)
        self.motion_inr = motion_inr or OracleINR(3, 2, encoding_config, mot_network_config, seed=seed, device=device)
        self.device = self.image_inr.device
        assert self.motion_inr.device == self.device
        self.masks = masks.to(self.device)
        self.num_movements, self.x, self.num_lines = masks.shape
        self.identy_grid = identity_grid(self.x, self.num_lines).to(self.device)
        self.input_grid = make_grids((self.num_movements, self.x, self.num_lines)).to(self.device)
        self._group_perm = None

    def redraw(self, rng: np.random.Generator):
        """New summation orders for the next step (see OracleINR.redraw) plus a new order of the motion groups,
        i.e. of the sum over groups in the backward of `repeat` (immoco.py:91)."""
        self.image_inr.redraw(rng)
        self.motion_inr.redraw(rng)
        self._group_perm = torch.from_numpy(rng.permutation(self.num_movements)).to(self.device)

    def forward(self):
        H, W, nM = self.x, self.num_lines, self.num_movements
        o = self.image_inr(self.identy_grid.view(-1, 2)).float().view(H, W, 2)
        image_prior = o[..., 0] + 1j * o[..., 1]
        images = image_prior.squeeze().unsqueeze(0).repeat(nM, 1, 1)
        grids = self.motion_inr.tanh(self.motion_inr(self.input_grid).float()).view(nM, H, W, 2) \
            + self.identy_grid.view(1, H, W, 2)
        masks = self.masks
        if self._group_perm is not None:      # same terms, another order of the group axis
            grids, masks = grids[self._group_perm], masks[self._group_perm]
        motion_images = torch.view_as_complex(
            F.grid_sample(torch.view_as_real(images).permute(0, 3, 1, 2), grids, mode="bilinear",
                          align_corners=False, padding_mode="zeros").permute(0, 2, 3, 1).contiguous())
        kspace_out = (FFT(image_prior).squeeze() * (1 - masks.sum(0)).float()) + (
            FFT(motion_images) * masks.float()).sum(0)
        return kspace_out, image_prior


def lambda_schedule(iters: int, lambda_ge: float) -> List[float]:
    """src/models/immoco.py:180-181 — lambda used AT iteration j (python float64).
    `if j % (iters // 10) and j > (iters // 2): lambda_ge *= 0.5` i.e. halves on
    every j > iters/2 that is NOT a multiple of iters//10."""
    step = iters // 10  # ZeroDivisionError for iters < 10, like the reference
    lam, out = float(lambda_ge), []
    for j in range(iters):
        out.append(lam)
        if j % step and j > (iters // 2):
            lam *= 0.5
    return out


def oracle_motion_correction(kspace_corr, masks, iters=200, learning_rate=1e-2, lambda_ge=1e-2,
                             seed=1337, model: Optional[OracleIMMoCo] = None, loss_hist: Optional[list] = None,
                             norm_scale: float = 16000.0, lambda_rule: str = "immoco", device=None,
                             on_forward=None):
    """src/models/immoco.py:116-206 on the CPU (no .cuda()) or, with `device` / a model built on one, on that
    device with ATen kernels (the device oracle).  norm_scale=8000 and lambda_rule="downstream" give the variant
    copy of src/test/test_immoco_downstream.py:150-152,188-189.  `on_forward(j, image_prior, loss)` is called
    after every forward with detached tensors (per-iteration PSNR records without a host synchronisation);
    `loss_hist` costs one synchronisation per iteration."""
    model = model or OracleIMMoCo(masks, seed=seed, device=device)
    kspace_corr = kspace_corr.to(model.device)
    scale = kspace_corr.abs().max()
    kspace_input = kspace_corr.div(scale).mul(norm_scale).clone().detach()
    opt = torch.optim.Adam([
        {"params": model.motion_inr.parameters(), "lr": learning_rate},
        {"params": model.image_inr.parameters(), "lr": learning_rate},
    ])
    for j in range(iters):
        opt.zero_grad()
        kfm, image_prior = model()
        loss = F.mse_loss(torch.view_as_real(kfm), torch.view_as_real(kspace_input)) \
            + gradient_entropy_loss(image_prior).mul(lambda_ge)
        loss.backward()
        opt.step()
        if on_forward is not None:
            on_forward(j, image_prior.detach(), loss.detach())
        if loss_hist is not None:
            loss_hist.append(float(loss.item()))
        if lambda_rule == "immoco":
            if j % (iters // 10) and j > (iters // 2):
                lambda_ge *= 0.5
        else:  # test_immoco_downstream.py:188-189
            if j % 10 == 0 and j > 80:
                lambda_ge *= 0.5
    return image_prior, kfm


# --------------------------------------------------------------------------
# metrics: src/utils/evaluate.py:19-47, crop src/test/test_immoco.py:77-81
# --------------------------------------------------------------------------
def normalize(x: torch.Tensor) -> torch.Tensor:
    """evaluate.py:19-29."""
    if x.shape[0] > 1:
        mx = x.view(x.shape[0], -1).max(1).values
        mn = x.view(x.shape[0], -1).min(1).values
        return (x - mn.view(-1, 1, 1, 1)) / ((mx - mn).view(-1, 1, 1, 1) + 1e-24)
    return (x - x.min()) / (x.max() - x.min() + 1e-24)


def rmse(x, y):
    """evaluate.py:32-34."""
    return torch.sqrt(torch.mean((x - y) ** 2))


def my_psnr(img1, img2, data_range=None, reduction="mean"):
    """evaluate.py:37-47."""
    mse = torch.mean((img1 - img2) ** 2, dim=(1, 2, 3))
    mp = img2.view(img2.shape[0], -1).max(1).values if data_range is None else data_range
    v = 20 * torch.log10(mp / torch.sqrt(mse))
    return v if reduction == "none" else v.mean()


def crop_psnr(pred_abs: torch.Tensor, gt_abs: torch.Tensor) -> float:
    """test_immoco.py:74-85 restricted to PSNR: centre-half crop, min-max normalise, data_range=1."""
    H, W = gt_abs.shape[-2:]
    c0, c1 = int(H / 4), int(W / 4)
    p = pred_abs[c0:-c0, c1:-c1][None, None]
    g = gt_abs[c0:-c0, c1:-c1][None, None]
    return float(my_psnr(normalize(p), normalize(g), data_range=1.0))


# --------------------------------------------------------------------------
# motion simulator: src/utils/motion_utils.py:7-34,112-202
# --------------------------------------------------------------------------
def generate_list(size, n_movements, mingap=4, acs=24):
    """motion_utils.py:7-24."""
    slack = size - mingap * (n_movements - 1)
    steps = torch.randint(0, slack, (1,))[0]
    inc = torch.hstack([torch.ones((steps,), dtype=torch.long), torch.zeros((n_movements,), dtype=torch.long)])
    inc = inc[torch.randperm(inc.shape[0])]
    locs = torch.argwhere(inc == 0).flatten()
    return torch.cumsum(inc, dim=0)[locs] + mingap * torch.arange(0, n_movements)


def get_rand_int(data_range, size=None):
    """motion_utils.py:27-34."""
    if size is None:
        r = torch.randint(data_range[0], data_range[1], size=(1,))
        if r == 0:
            r = r + 1
    else:
        r = torch.randint(data_range[0], data_range[1], size=size)
    return r


def motion_simulation2D(image_2d: torch.Tensor, n_movements: Optional[int] = None):
    """motion_utils.py:121-202 (same RNG call order so a shared torch seed gives
    the same corruption)."""
    ksp = FFT(image_2d)
    x, num_lines = ksp.shape
    if n_movements is None:
        n_movements = get_rand_int([5, 20]).item()
    mingap = num_lines // n_movements
    acs = int(num_lines * 0.08)
    rand_list = generate_list(num_lines, n_movements, mingap, acs)
    mask = torch.zeros((x, num_lines), dtype=torch.long)
    rotations = torch.zeros((n_movements,))
    translations = torch.zeros((n_movements, 2))
    for motion in range(n_movements):
        shift = [get_rand_int([-10, 10]).item(), get_rand_int([-10, 10]).item()]
        angle = get_rand_int([-10, 10])
        a = torch.deg2rad(angle)
        rot = torch.tensor([[torch.cos(a), -torch.sin(a)], [torch.sin(a), torch.cos(a)]])
        aff = torch.tensor([[1, 0, shift[0]], [0, 1, shift[1]]]).float()
        aff[:2, :2] = rot
        aff = aff.view(1, 2, 3)
        aff[:, :, -1] /= (torch.tensor(image_2d[0, ...].shape) * 2.0) - 1
        grid = F.affine_grid(aff, (1, 1, x, num_lines), align_corners=True)
        re = F.grid_sample(image_2d[None, None].real, grid.float(), mode="bilinear",
                           padding_mode="border", align_corners=False)
        im = F.grid_sample(image_2d[None, None].imag, grid.float(), mode="bilinear",
                           padding_mode="border", align_corners=False)
        ksp_m = FFT(re + 1j * im).squeeze()
        w0 = rand_list[motion]
        w1 = w0 + get_rand_int([1, 10])
        ksp[..., w0:w1] = ksp_m[..., w0:w1]
        mask[:, w0:w1] = 1
        rotations[motion] = angle
        translations[motion, :] = torch.tensor(shift)
    return ksp, mask, rotations, translations
