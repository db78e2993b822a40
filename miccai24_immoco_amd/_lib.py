"""ctypes binding of libimmoco_hip.so (C-ABI declared in include/immoco_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` /
``make -C miccai24_immoco_amd/csrc``.  Loading is lazy so that the package can
be imported on a machine without the library for host-only logic, but every
compute call goes through :func:`lib` and raises loudly if it is missing.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# IMMOCO_LIB_PATH: load another build of the same C-ABI (the diagnostics build `make -C csrc diag`, whose IMMOCO_*
# A/B switches the shipped library does not contain)
LIB_PATH = os.environ.get("IMMOCO_LIB_PATH") or os.path.join(_HERE, "csrc", "libimmoco_hip.so")
_lock = threading.Lock()
_lib = None

MAX_LEVELS = 16
ACT_RELU, ACT_TANH = 0, 1


class GridCfg(C.Structure):
    _fields_ = [("dims", C.c_int32), ("n_levels", C.c_int32), ("n_features", C.c_int32),
                ("log2_hashmap_size", C.c_int32), ("base_resolution", C.c_int32),
                ("per_level_scale", C.c_float)]


class GridGeometry(C.Structure):
    _fields_ = [("offset", C.c_uint32 * (MAX_LEVELS + 1)), ("resolution", C.c_uint32 * MAX_LEVELS),
                ("size", C.c_uint32 * MAX_LEVELS), ("scale", C.c_float * MAX_LEVELS),
                ("hashed", C.c_uint8 * MAX_LEVELS)]


class MlpCfg(C.Structure):
    _fields_ = [("n_in", C.c_int32), ("n_hidden", C.c_int32), ("n_out", C.c_int32),
                ("n_out_padded", C.c_int32), ("activation", C.c_int32)]


class SolverCfg(C.Structure):
    _fields_ = [("H", C.c_int32), ("W", C.c_int32), ("nM", C.c_int32),
                ("image_grid", GridCfg), ("motion_grid", GridCfg),
                ("image_mlp", MlpCfg), ("motion_mlp", MlpCfg),
                ("use_graph", C.c_int32), ("atomic_scatter", C.c_int32), ("grad_parts", C.c_int32),
                ("serial_chains", C.c_int32), ("table_fp16", C.c_int32),
                ("batch_lanes", C.c_int32), ("mlp_fp16", C.c_int32), ("batch_pair", C.c_int32)]


_P, _I32, _I64, _F, _U32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_uint32
_GP, _MP = C.POINTER(GridCfg), C.POINTER(MlpCfg)

# name -> (restype, argtypes); must list every function declared in include/immoco_hip.h
PROTOTYPES = {
    "immoco_version": (C.c_int, []),
    "immoco_last_error": (C.c_int, [C.c_char_p, C.c_size_t]),
    "immoco_grid_geometry_query": (C.c_int, [_GP, C.POINTER(GridGeometry)]),
    "immoco_hashgrid_fwd": (C.c_int, [_GP, _P, _I64, _P, _P, _I64, _I64, _P]),
    "immoco_hashgrid_fwd_f16": (C.c_int, [_GP, _P, _I64, _P, _P, _I64, _I64, _P]),
    "immoco_hashgrid_fwd_lattice": (C.c_int, [_GP, _I32, _I32, _I32, _P, _P, _P, _P, _P, _I64, _I64, _P]),
    "immoco_hashgrid_bwd": (C.c_int, [_GP, _P, _I64, _P, _I64, _I64, _P, _P]),
    "immoco_mlp_fwd": (C.c_int, [_MP, _P, _I64, _I64, _I64, _P, _P, _P, _P]),
    "immoco_mlp_bwd": (C.c_int, [_MP, _P, _I64, _I64, _I64, _P, _P, _P, _P, _P, _P, _P]),
    "immoco_mlp_bwd_split": (C.c_int, [_MP, _P, _I64, _I64, _I64, _P, _P, _P, _P, _P, _P, _P]),
    "immoco_mlp_fwd_half": (C.c_int, [_MP, _P, _I64, _I64, _I64, _P, _P, _P, _P]),
    "immoco_mlp_bwd_half": (C.c_int, [_MP, _P, _I64, _I64, _I64, _P, _P, _P, _F, _P, _P, _P, _P]),
    "immoco_mlp_fwd_bf16x2": (C.c_int, [_MP, _P, _I64, _I64, _I64, _P, _P, _P, _P]),
    "immoco_mlp_bwd_bf16x2": (C.c_int, [_MP, _P, _I64, _I64, _I64, _P, _P, _P, _P, _P, _P, _P]),
    "immoco_init_params": (C.c_int, [_GP, _MP, _U32, _P, _P]),
    "immoco_warp_fwd": (C.c_int, [_P, _P, _I32, _I32, _I32, _P, _P]),
    "immoco_warp_bwd": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, _P, _P, _P]),
    "immoco_affine_warp_border": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _P, _P]),
    "immoco_band_replace": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _P, _P, _P]),
    "immoco_affine_bicubic_fwd": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _P, _P]),
    "immoco_affine_bicubic_bwd": (C.c_int, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _P, _P]),
    "immoco_fft2c": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _P]),
    "immoco_kspace_select": (C.c_int, [_P, _P, _I32, _I32, _I32, _P, _P]),
    "immoco_dc_loss": (C.c_int, [_P, _P, _I32, _I32, _P, _P, _P]),
    "immoco_ge_loss": (C.c_int, [_P, _I32, _I32, _F, _P, _P, _P]),
    "immoco_adam_step": (C.c_int, [_P, _P, _P, _P, _I64, _F, _F, _F, _F, _I32, _P]),
    "immoco_extract_movement_groups": (C.c_int, [_P, _I32, _P, _P, _P]),
    "immoco_groups_to_matrix": (C.c_int, [_P, _I32, _I32, _P, _P]),
    "immoco_groups_to_masks": (C.c_int, [_P, _I32, _I32, _I32, _P, _P]),
    "immoco_masks_to_groups": (C.c_int, [_P, _I32, _I32, _I32, _P, _P]),
    "immoco_normalize_kspace": (C.c_int, [_P, _I64, _F, _P, _P, _P]),
    "immoco_solver_create": (C.c_int, [C.POINTER(SolverCfg), C.POINTER(C.c_void_p)]),
    "immoco_solver_destroy": (C.c_int, [_P]),
    "immoco_solver_workspace_bytes": (C.c_int64, [_P]),
    "immoco_solver_n_params": (C.c_int64, [_P, _I32]),
    "immoco_solver_set_lattice": (C.c_int, [_P, _P, _P, _P, _P]),
    "immoco_solver_solve": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I32, _F,
                                      C.POINTER(C.c_float), _I32, _P, _P, _P, _P]),
    "immoco_solver_solve_batch": (C.c_int, [_P, _I32, _P, _P, _P, _P, _P, _P, _I32, _F,
                                            C.POINTER(C.c_float), _I32, _P, _P, _P, _P]),
    "immoco_solver_forward": (C.c_int, [_P, _P, _P, _P, _P, _P, _P]),
    "immoco_solver_profile": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I32, _F, _F, _P]),
    "immoco_grid_plan_create": (C.c_int, [_GP, _I32, _I32, _I32, _P, _P, _P, C.POINTER(C.c_void_p), _P]),
    "immoco_grid_plan_destroy": (C.c_int, [_P]),
    "immoco_grid_plan_bytes": (C.c_int64, [_P]),
    "immoco_grid_plan_bwd": (C.c_int, [_P, _P, _P, _P]),
    "immoco_solver_phase_times": (C.c_int, [_P, C.POINTER(C.c_char_p), C.POINTER(C.c_float), _I32]),
    "immoco_solver_graph_active": (C.c_int, [_P]),
    "immoco_solver_dominant_kernel_ms": (C.c_float, [_P]),
    "immoco_solver_set_graph": (C.c_int, [_P, _I32]),
    "immoco_solver_plan_entries": (C.c_int64, [_P, _I32]),
    "immoco_probe_gather": (C.c_int, [_I64, _I32, _I64, _I32, _I32, _P, C.POINTER(C.c_float)]),
}


class ImmocoError(RuntimeError):
    pass


def lib_available() -> bool:
    return os.path.exists(LIB_PATH)


def load(path: str = LIB_PATH) -> C.CDLL:
    """dlopen the library and bind every prototype (no GPU needed)."""
    if not os.path.exists(path):
        raise ImmocoError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C miccai24_immoco_amd/csrc`. There is no CPU/PyTorch fallback.")
    h = C.CDLL(path, mode=C.RTLD_GLOBAL)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(h, name)
        fn.restype, fn.argtypes = res, args
    return h


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                _lib = load()
    return _lib


def last_error() -> str:
    buf = C.create_string_buffer(512)
    lib().immoco_last_error(buf, 512)
    return buf.value.decode("utf-8", "replace")


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise ImmocoError(f"{what or 'immoco call'} failed (rc={rc}): {last_error()}")


def stream_ptr() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def require_gpu(*tensors, what="operator"):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise ImmocoError(f"{what}: tensor is on {t.device}; this package only computes on the GPU "
                              "through libimmoco_hip.so (no CPU fallback)")


def grid_cfg(n_dims: int, enc: dict) -> GridCfg:
    if str(enc.get("otype", "Grid")).lower() not in ("grid", "hashgrid"):
        raise ImmocoError(f"unsupported encoding otype {enc.get('otype')!r}")
    if str(enc.get("type", "Hash")).lower() != "hash":
        raise ImmocoError(f"unsupported grid type {enc.get('type')!r}")
    if str(enc.get("interpolation", "Linear")).lower() != "linear":
        raise ImmocoError(f"unsupported interpolation {enc.get('interpolation')!r}")
    # "fine_resolution" (reference immoco.py:34) is not a tiny-cuda-nn key: ignored, like upstream
    return GridCfg(n_dims, int(enc.get("n_levels", 16)), int(enc.get("n_features_per_level", 2)),
                   int(enc.get("log2_hashmap_size", 19)), int(enc.get("base_resolution", 16)),
                   float(enc.get("per_level_scale", 2.0)))


def mlp_cfg(n_in: int, n_out: int, net: dict) -> MlpCfg:
    otype = str(net.get("otype", "FullyFusedMLP")).lower()
    if otype not in ("fullyfusedmlp", "cutlassmlp"):
        raise ImmocoError(f"unsupported network otype {net.get('otype')!r}")
    if int(net.get("n_hidden_layers", 1)) != 1:
        raise ImmocoError("only n_hidden_layers=1 is supported (the reference's configs)")
    if str(net.get("output_activation", "None")).lower() != "none":
        raise ImmocoError("only output_activation=None is supported")
    act = {"relu": ACT_RELU, "tanh": ACT_TANH}.get(str(net.get("activation", "ReLU")).lower())
    if act is None:
        raise ImmocoError(f"unsupported activation {net.get('activation')!r}")
    pad = 16 if otype == "fullyfusedmlp" else 8
    return MlpCfg(n_in, int(net.get("n_neurons", 64)), n_out, (n_out + pad - 1) // pad * pad, act)


def geometry(cfg: GridCfg) -> GridGeometry:
    g = GridGeometry()
    check(lib().immoco_grid_geometry_query(C.byref(cfg), C.byref(g)), "grid_geometry_query")
    return g
