"""CPU-side checks: the C-ABI library builds, loads and exports every symbol that
include/immoco_hip.h declares; host-only entry points (geometry, argument
validation) agree with the oracle; host logic of the Python mirror."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from miccai24_immoco_amd import _lib
    if not _lib.lib_available():
        import __graft_entry__ as ge
        ge.build()
    _lib.lib()
    return _lib


def declared_functions():
    src = open(os.path.join(ROOT, "include", "immoco_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(immoco_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound(L):
    names = declared_functions()
    assert len(names) >= 25
    h = L.lib()
    for n in names:
        assert hasattr(h, n), f"{n} declared in include/immoco_hip.h but not exported"
        assert n in L.PROTOTYPES, f"{n} has no ctypes prototype"
    assert set(L.PROTOTYPES) == set(names)
    assert h.immoco_version() >= 100


def test_geometry_query_matches_oracle(L):
    from oracle import immoco_oracle as orc
    for dims in (2, 3):
        for enc in (orc.encoding_config, dict(orc.encoding_config, per_level_scale=1.5, base_resolution=12,
                                              log2_hashmap_size=15, n_levels=12)):
            geo = orc.geometry_from_config(dims, enc)
            g = L.geometry(L.grid_cfg(dims, enc))
            nl = geo.n_levels
            assert list(g.offset)[: nl + 1] == geo.offsets
            assert list(g.resolution)[:nl] == geo.resolutions and list(g.size)[:nl] == geo.sizes
            assert [bool(x) for x in list(g.hashed)[:nl]] == geo.hashed
            if enc is orc.encoding_config:      # the reference's config: exact (powers of two)
                assert [float(x) for x in list(g.scale)[:nl]] == geo.scales
            else:                               # exp2f/log2f are libm-dependent to 1 ulp off powers of two
                np.testing.assert_allclose([float(x) for x in list(g.scale)[:nl]], geo.scales, rtol=3e-7)


def test_invalid_arguments_return_errors(L):
    h = L.lib()
    bad = L.GridCfg(4, 16, 2, 19, 16, 2.0)
    g = L.GridGeometry()
    assert h.immoco_grid_geometry_query(C.byref(bad), C.byref(g)) == -1
    assert "dims" in L.last_error()
    with pytest.raises(L.ImmocoError):
        L.check(h.immoco_fft2c(None, None, 1, 0, 4, 0, None), "fft2c")
    with pytest.raises(L.ImmocoError):
        L.mlp_cfg(32, 2, {"otype": "FullyFusedMLP", "activation": "Sine", "n_neurons": 64})
    with pytest.raises(L.ImmocoError):
        L.grid_cfg(2, {"otype": "Frequency"})
    cfg = L.SolverCfg(321, 320, 2, L.grid_cfg(2, {}), L.grid_cfg(3, {}), L.mlp_cfg(32, 2, {"otype": "CutlassMLP", "n_neurons": 256}),
                      L.mlp_cfg(32, 2, {"otype": "FullyFusedMLP", "activation": "Tanh", "n_neurons": 64}), 1)
    out = C.c_void_p()
    assert h.immoco_solver_create(C.byref(cfg), C.byref(out)) == -1 and "even" in L.last_error()


def test_cpu_tensors_are_refused(L):
    import miccai24_immoco_amd as pkg
    with pytest.raises(L.ImmocoError):
        pkg.FFT(torch.zeros(4, 4, dtype=torch.complex64))
    with pytest.raises(L.ImmocoError):
        pkg.GradientEntropyLoss()(torch.zeros(4, 4, dtype=torch.complex64))
    with pytest.raises(L.ImmocoError):
        pkg.extract_movement_groups(torch.zeros(8, dtype=torch.bool))
    with pytest.raises(L.ImmocoError):
        pkg.imcoco_motion_correction(torch.zeros(16, 16, dtype=torch.complex64), torch.zeros(1, 16, 16, dtype=torch.long))


def test_missing_library_fails_loudly(L, tmp_path):
    with pytest.raises(L.ImmocoError):
        L.load(str(tmp_path / "libimmoco_hip.so"))


def test_lambda_schedule_host_logic():
    from miccai24_immoco_amd.models.immoco import lambda_schedule
    from oracle import immoco_oracle as orc
    for iters in (10, 50, 200, 3000):
        assert lambda_schedule(iters, 1e-2) == orc.lambda_schedule(iters, 1e-2)
    with pytest.raises(ZeroDivisionError):
        lambda_schedule(9, 1e-2)
    d = lambda_schedule(200, 1e-2, rule="downstream")
    assert d[91] == 0.5e-2 and d[90] == 1e-2 and d[101] == 0.25e-2


def test_make_grids_matches_golden(golden):
    from miccai24_immoco_amd.models.immoco import make_grids
    g = golden("ops")
    assert np.array_equal(make_grids((2, 3, 4)).numpy(), g["make_grids_2_3_4"])
    assert np.array_equal(make_grids((1, 3, 5)).numpy(), g["make_grids_1_3_5"])


def test_synth_cpu_slice_and_gpu_only_package_generator():
    """oracle/synth_cpu.make_slice (oracle motion simulator, pinned by test_oracle_golden.py) is seeded and
    self-consistent; the package's generator refuses CPU devices (no CPU path in the product)."""
    from miccai24_immoco_amd import synth
    from miccai24_immoco_amd._lib import ImmocoError
    from oracle import synth_cpu
    a, b = synth_cpu.make_slice(32, 32, 3, 4), synth_cpu.make_slice(32, 32, 3, 4)
    assert torch.equal(a["kspace"], b["kspace"]) and torch.equal(a["lines"], b["lines"])
    assert a["kspace"].dtype == torch.complex64 and a["lines"].dtype == torch.bool and int(a["lines"].sum()) > 0
    assert torch.equal(a["gt"], synth.phantom(32, 32, 1004))
    with pytest.raises(ImmocoError):
        synth.make_slice(32, 32, 3, 4, device="cpu")


def test_header_is_plain_c_and_a_c_host_links(tmp_path):
    """include/immoco_hip.h compiles as C99 without any HIP header ("a C host includes the
    header directly"), and a C program linked against libimmoco_hip.so calls the GPU-free entry points."""
    import subprocess
    from miccai24_immoco_amd import _lib as L
    inc = os.path.join(ROOT, "include")
    src = os.path.join(ROOT, "tests", "c_abi_smoke.c")
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-I" + inc,
                        "-x", "c", os.path.join(inc, "immoco_hip.h")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    if not L.lib_available():
        pytest.fail("libimmoco_hip.so is not built (run __graft_entry__.build())")
    libdir = os.path.dirname(L.LIB_PATH)
    exe = str(tmp_path / "c_abi_smoke")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I" + inc, src, "-o", exe, "-L" + libdir,
                        "-limmoco_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert r.stdout.startswith("entries 7114752 16:0 32:0 64:0 128:1")
    assert r.stdout.strip().endswith("65536:0 131072:0 262144:0 524288:0")   # wrapped-stride levels: not hashed


def test_shipped_library_reads_no_environment_switch():
    """the result-changing / A-B switches (IMMOCO_CSR_STREAM, IMMOCO_MLP_IMPL, ...) exist only in the
    diagnostics build (`make -C miccai24_immoco_amd/csrc diag`, -DIMMOCO_DIAG); the shipped library holds none of their
    names."""
    from miccai24_immoco_amd import _lib as L
    if not L.lib_available():
        pytest.fail("libimmoco_hip.so is not built (run __graft_entry__.build())")
    blob = open(L.LIB_PATH, "rb").read()
    assert b"IMMOCO_CSR_STREAM" not in blob and b"IMMOCO_MLP_IMPL" not in blob and b"IMMOCO_" not in blob


def test_solver_cfg_fields_are_validated(L):
    """mlp_fp16 / batch_pair / serial_chains were `reserved` words before round 3; garbage is refused."""
    h = L.lib()
    def cfg(**kw):
        base = dict(use_graph=1, atomic_scatter=0, grad_parts=0, serial_chains=2, table_fp16=0, batch_lanes=0, mlp_fp16=0, batch_pair=0)
        base.update(kw)
        return L.SolverCfg(32, 32, 2, L.grid_cfg(2, {}), L.grid_cfg(3, {}), L.mlp_cfg(32, 2, {"otype": "CutlassMLP", "n_neurons": 256}),
                           L.mlp_cfg(32, 2, {"otype": "FullyFusedMLP", "activation": "Tanh", "n_neurons": 64}),
                           base["use_graph"], base["atomic_scatter"], base["grad_parts"], base["serial_chains"], base["table_fp16"],
                           base["batch_lanes"], base["mlp_fp16"], base["batch_pair"])
    for bad, word in ((dict(mlp_fp16=3), "mlp_fp16"), (dict(mlp_fp16=-1), "mlp_fp16"), (dict(batch_pair=7), "batch_pair"),
                      (dict(serial_chains=5), "serial_chains")):
        out = C.c_void_p()
        c = cfg(**bad)
        assert h.immoco_solver_create(C.byref(c), C.byref(out)) == -1 and word in L.last_error(), (bad, L.last_error())
