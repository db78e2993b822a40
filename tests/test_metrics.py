"""On-device metrics: the torch restatements of piq's ssim / haarpsi against the
independent float64 scipy oracle and known answers.  (piq itself is absent: parity with it is unpinned.)"""
import numpy as np
import pytest
import torch

from miccai24_immoco_amd.utils import evaluate as E
from oracle import metrics_oracle as MO


def _pair(H, W, seed, noise=0.08):
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
    a = (torch.exp(-(xx ** 2 + yy ** 2) * 3) + 0.3 * torch.sin(9 * xx) * torch.cos(7 * yy)).clamp(0, None)
    a = a / a.max()
    b = (a + noise * torch.randn(H, W, generator=g)).clamp(0, 1)
    return a[None, None].contiguous(), b[None, None].contiguous()


@pytest.mark.parametrize("shape", [(160, 160), (64, 48), (33, 21), (400, 390)])
def test_ssim_vs_scipy_oracle(shape):
    a, b = _pair(*shape, seed=1)
    got = float(E.ssim(a, b, data_range=1.0, kernel_size=11))
    ref = MO.ssim_np(a[0, 0].numpy(), b[0, 0].numpy())
    assert abs(got - ref) <= 2e-5, (got, ref)
    assert abs(float(E.ssim(a, a)) - 1.0) <= 1e-6
    assert 0.0 < got < 1.0


@pytest.mark.parametrize("shape", [(160, 160), (64, 48), (33, 21)])
def test_haarpsi_vs_scipy_oracle(shape):
    a, b = _pair(*shape, seed=2)
    got = float(E.haarpsi(a, b, scales=3))
    ref = MO.haarpsi_np(a[0, 0].numpy(), b[0, 0].numpy())
    assert abs(got - ref) <= 1e-4 * max(1.0, ref), (got, ref)
    assert abs(float(E.haarpsi(a, a)) - 1.0) <= 1e-4
    assert abs(float(E.haarpsi(b, a)) - got) <= 1e-6          # symmetric
    worse = float(E.haarpsi(a, _pair(*shape, seed=2, noise=0.3)[1]))
    assert worse < got < 1.0                                  # monotone in the distortion


def test_metric_batches_and_errors():
    a, b = _pair(40, 40, seed=3)
    a2, b2 = torch.cat([a, b]), torch.cat([b, b])
    v = E.ssim(a2, b2, reduction="none")
    assert v.shape == (2,) and abs(float(v[1]) - 1.0) <= 1e-6
    assert abs(float(E.ssim(a2, b2)) - float(v.mean())) <= 1e-7
    h = E.haarpsi(a2, b2, reduction="none")
    assert h.shape == (2,) and abs(float(h[0]) - float(E.haarpsi(a, b))) <= 1e-6
    with pytest.raises(ValueError):
        E.ssim(a, b, kernel_size=10)
    with pytest.raises(ValueError):
        E.ssim(a[..., :8, :8], b[..., :8, :8])
    with pytest.raises(ValueError):
        E.ssim(a * 2, b)
    with pytest.raises(ValueError):
        E.haarpsi(a[..., :8, :8], b[..., :8, :8])
    with pytest.raises(ValueError):
        E.calmetric2D(a[0], b[0])
    with pytest.raises(ValueError):
        E.calmetric2D(a[..., :9, :9], b[..., :9, :9])


def test_calmetric2d_and_3d_consistency():
    a, b = _pair(64, 64, seed=4)
    ps, ss, hp, rm = E.calmetric2D(b * 3.0 + 1.0, a)           # normalisation removes scale and offset
    na, nb = E.normalize(a), E.normalize(b)
    assert abs(float(ps) - float(E.my_psnr(nb, na, data_range=1.0))) <= 1e-5
    assert abs(float(ss) - MO.ssim_np(nb[0, 0].numpy(), na[0, 0].numpy())) <= 2e-5
    assert abs(float(hp) - MO.haarpsi_np(nb[0, 0].numpy(), na[0, 0].numpy())) <= 1e-4
    assert abs(float(rm) - float(E.rmse(nb, na))) <= 1e-7
    c, d = _pair(64, 64, seed=5, noise=0.2)
    m3 = E.calmetric3D(torch.cat([b, d]), torch.cat([a, c]))
    m2a, m2b = E.calmetric2D(b, a), E.calmetric2D(d, c)
    for k in range(4):
        assert abs(float(m3[k]) - 0.5 * (float(m2a[k]) + float(m2b[k]))) <= 1e-5
    rec = E.slice_metrics(torch.complex(b[0, 0], torch.zeros(64, 64)), torch.complex(a[0, 0], torch.zeros(64, 64)))
    assert set(rec) == {"ssim", "psnr", "haar_psi", "rmse"}
    p2 = E.crop_psnr(b[0, 0], a[0, 0])
    assert abs(float(rec["psnr"]) - p2) <= 1e-5


def _check_product_metrics_vs_reference_golden(g, dev):
    """The PRODUCT functions behind bench.py's psnr_db (utils/evaluate.py: normalize / my_psnr / rmse /
    crop_psnr) against vectors produced by the reference's own src/utils/evaluate.py:19-47
    (tools/gen_golden.py -> tests/golden/ops.npz)."""
    a, b = torch.from_numpy(g["metric_a"]).to(dev), torch.from_numpy(g["metric_b"]).to(dev)
    na, nb = E.normalize(a), E.normalize(b)
    np.testing.assert_allclose(na.cpu().numpy(), g["normalize_a"], rtol=1e-6, atol=1e-7)
    assert abs(float(E.my_psnr(na, nb, data_range=1.0)) - float(g["psnr_ab"])) <= 1e-4
    assert abs(float(E.rmse(na, nb)) - float(g["rmse_ab"])) <= 1e-6
    # batch branch of normalize (evaluate.py:20-26) and data_range=None (peak of img2)
    ab = torch.cat([a, b * 2 + 1])
    nab = E.normalize(ab)
    np.testing.assert_allclose(nab[0:1].cpu().numpy(), g["normalize_a"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(nab[1:2].cpu().numpy(), nb.cpu().numpy(), rtol=1e-5, atol=1e-6)
    assert abs(float(E.my_psnr(na, nb)) - float(g["psnr_ab"]) - 20 * np.log10(float(nb.max()))) <= 1e-4
    # crop_psnr == the same chain on the centre-half crop (test_immoco.py:77-81)
    x, y = a[0, 0], b[0, 0]
    c = x.shape[0] // 4
    exp = float(E.my_psnr(E.normalize(x[c:-c, c:-c][None, None]), E.normalize(y[c:-c, c:-c][None, None]), data_range=1.0))
    assert abs(E.crop_psnr(x, y) - exp) <= 1e-6


def test_product_psnr_rmse_normalize_vs_reference_golden(golden):
    _check_product_metrics_vs_reference_golden(golden("ops"), "cpu")


@pytest.mark.gpu
def test_product_psnr_rmse_normalize_vs_reference_golden_on_device(golden):
    _check_product_metrics_vs_reference_golden(golden("ops"), "cuda")


@pytest.mark.gpu
def test_metrics_on_device_match_cpu():
    a, b = _pair(160, 160, seed=6)
    cpu = [float(v) for v in E.calmetric2D(b, a)]
    dev = E.calmetric2D(b.cuda(), a.cuda())
    assert all(v.is_cuda for v in dev)
    np.testing.assert_allclose([float(v) for v in dev], cpu, rtol=2e-4, atol=1e-5)
