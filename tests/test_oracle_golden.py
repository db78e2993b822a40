"""Oracle (oracle/immoco_oracle.py) vs golden vectors produced by the REFERENCE's
own Python (tools/gen_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import immoco_oracle as orc
from conftest import expand_masks


def c(x):
    return torch.from_numpy(x)


@pytest.mark.parametrize("tag", ["even", "mod2", "odd"])
def test_fft_ifft(golden, tag):
    g = golden("ops")
    x = c(g[f"fft_{tag}_in"])
    np.testing.assert_allclose(orc.FFT(x).numpy(), g[f"fft_{tag}_out"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(orc.IFFT(x).numpy(), g[f"ifft_{tag}_out"], rtol=1e-5, atol=1e-6)


def test_gradient_entropy(golden):
    g = golden("ops")
    x = c(g["ge_in"]).clone().requires_grad_(True)
    loss = orc.gradient_entropy_loss(x)
    loss.backward()
    assert abs(float(loss.detach()) - float(g["ge_loss"])) <= 1e-4 * abs(float(g["ge_loss"]))
    np.testing.assert_allclose(x.grad.numpy(), g["ge_grad"], rtol=1e-5, atol=1e-5)
    assert np.isfinite(x.grad.numpy().view(np.float32)).all()


def test_make_grids(golden):
    g = golden("ops")
    assert np.array_equal(orc.make_grids((2, 3, 4)).numpy(), g["make_grids_2_3_4"])
    assert np.array_equal(orc.make_grids((1, 3, 5)).numpy(), g["make_grids_1_3_5"])


def test_metrics(golden):
    g = golden("ops")
    a, b = c(g["metric_a"]), c(g["metric_b"])
    np.testing.assert_allclose(orc.normalize(a).numpy(), g["normalize_a"], rtol=1e-6)
    assert abs(float(orc.my_psnr(orc.normalize(a), orc.normalize(b), data_range=1.0)) - float(g["psnr_ab"])) < 1e-4
    assert abs(float(orc.rmse(orc.normalize(a), orc.normalize(b))) - float(g["rmse_ab"])) < 1e-6


@pytest.mark.parametrize("tag", ["typical", "last_true", "first_true", "single", "all_true",
                                 "alternating", "random320"])
def test_extract_movement_groups_bit_exact(golden, tag):
    g = golden("masks")
    v = c(g[f"{tag}_vec"]).bool()
    groups = orc.extract_movement_groups(v, make_list=False)
    assert groups.dtype == torch.int64 and groups.shape == (len(v), len(v))
    assert np.array_equal(groups[0].numpy().astype(np.int32), g[f"{tag}_groups"])
    assert bool((groups == groups[:1]).all())
    ml = orc.extract_movement_groups(v, make_list=True)
    assert tuple(ml.shape) == tuple(g[f"{tag}_list_shape"])
    assert ml.dtype == torch.int64
    assert np.array_equal(ml[:, 0, :].numpy().astype(np.uint8), g[f"{tag}_list_row0"])
    assert bool((ml == ml[:, :1, :]).all())
    # compact column->group form used by the kernels
    cg = orc.col_group_from_masks(ml).numpy()
    assert np.array_equal(cg, g[f"{tag}_groups"])


def test_extract_movement_groups_empty():
    ml = orc.extract_movement_groups(torch.zeros(8, dtype=torch.bool), make_list=True)
    assert tuple(ml.shape) == (0, 8, 8)


@pytest.mark.parametrize("tag", ["s32", "s64"])
def test_motion_simulation(golden, tag):
    g = golden("motion_sim")
    torch.manual_seed(int(g[f"{tag}_seed"]))
    ksp, mask, rot, tr = orc.motion_simulation2D(c(g[f"{tag}_img"]).clone(), n_movements=int(g[f"{tag}_nm"]))
    assert np.array_equal(mask[0].numpy().astype(np.uint8), g[f"{tag}_mask_row0"])   # bit-exact lines
    assert np.array_equal(rot.numpy(), g[f"{tag}_rot"]) and np.array_equal(tr.numpy(), g[f"{tag}_tr"])
    np.testing.assert_allclose(ksp.numpy(), g[f"{tag}_ksp"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("tag", ["c32", "c48"])
def test_forward_operator(golden, tag):
    g = golden("solver")
    H = g[f"{tag}_gt"].shape[0]
    masks = expand_masks(g[f"{tag}_masks_row0"], H)
    model = orc.OracleIMMoCo(masks, image_inr=orc.OracleINR(2, 2, orc.encoding_config, orc.network_config),
                             motion_inr=orc.OracleINR(3, 2, orc.encoding_config, orc.mot_network_config))
    assert np.array_equal(model.identy_grid.numpy(), g[f"{tag}_identy_grid"])
    assert np.array_equal(model.input_grid.numpy(), g[f"{tag}_input_grid"])
    # identity grid is exactly the per-axis linspace (what the kernels consume)
    assert np.array_equal(model.identy_grid[0, 0, :, 0].numpy(), torch.linspace(-1, 1, H).numpy())
    assert np.array_equal(model.identy_grid[0, :, 0, 1].numpy(), torch.linspace(-1, 1, H).numpy())
    with torch.no_grad():
        k0, im0 = model()
    np.testing.assert_allclose(im0.numpy(), g[f"{tag}_fwd0_image"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(k0.numpy(), g[f"{tag}_fwd0_kspace"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("tag", ["c32", "c48"])
def test_solver_loop(golden, tag):
    """Reference loop (immoco.py:116-206) vs the oracle's restatement, same INR, same seeds."""
    g = golden("solver")
    H = g[f"{tag}_gt"].shape[0]
    masks = expand_masks(g[f"{tag}_masks_row0"], H)
    model = orc.OracleIMMoCo(masks, image_inr=orc.OracleINR(2, 2, orc.encoding_config, orc.network_config),
                             motion_inr=orc.OracleINR(3, 2, orc.encoding_config, orc.mot_network_config))
    img, kfm = orc.oracle_motion_correction(c(g[f"{tag}_ksp"]), masks, iters=int(g[f"{tag}_iters"]),
                                            learning_rate=1e-2, lambda_ge=1e-2, model=model)
    # The trajectory is chaotic in its details: Adam turns rounding-level gradient
    # differences of barely-touched hash entries into lr-sized steps, and torch's CPU
    # scatter-add order is itself nondeterministic (run-to-run max-rel diff ~1e-2 after
    # 20 iterations, relative L2 between identical runs 1e-3..4e-2, PSNR spread 0.005 dB;
    # measured).  Parity is therefore asserted as PSNR delta <= 0.1 dB (the north-star
    # tolerance) plus a loose relative-L2 bound.
    ref = g[f"{tag}_image_prior"]
    err = np.linalg.norm(img.detach().numpy() - ref) / np.linalg.norm(ref)
    assert err < 0.15, err
    errk = np.linalg.norm(kfm.detach().numpy() - g[f"{tag}_kfm"]) / np.linalg.norm(g[f"{tag}_kfm"])
    assert errk < 0.15, errk
    gt = c(g[f"{tag}_gt"]).abs()
    d = orc.crop_psnr(img.detach().abs(), gt) - orc.crop_psnr(c(np.abs(ref)), gt)
    assert abs(d) <= 0.1, d


def test_lambda_schedule():
    """immoco.py:180-181 quirk: 20 halvings @50 it, 95 @200, underflow to 0 @3000."""
    s = orc.lambda_schedule(50, 1e-2)
    assert s[0] == 1e-2 and s[26] == 1e-2 and s[27] == 0.5e-2
    final = s[-1] * (0.5 if (49 % 5 and 49 > 25) else 1.0)
    assert abs(final - 1e-2 * 0.5 ** 20) < 1e-20
    s200 = orc.lambda_schedule(200, 1e-2)
    assert sum(1 for a, b in zip(s200[:-1], s200[1:]) if b != a) + (1 if (199 % 20 and 199 > 100) else 0) == 95
    s3000 = orc.lambda_schedule(3000, 1e-2)
    assert s3000[-1] == 0.0
    with pytest.raises(ZeroDivisionError):
        orc.lambda_schedule(9, 1e-2)
