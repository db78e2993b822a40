"""The oracle is a FAMILY at the reference's 200-iteration setting.

`OracleINR(mlp_splitk=c)` sums every MLP product over c interleaved slices of its inner dimension: an equally valid fp32
evaluation order of the same sums (what another GEMM tiling does), equal to the plain oracle to ~1e-7 per step.  The
committed device-oracle draws (tests/golden/c2_device_oracle_draws.npz, tools/device_oracle_sampler.py --mlp-splitk) show
what that does to the statistic the 200-iteration parity cells use: the spread of the oracle's draws comes from the order of
its fp32 atomics alone (3e-8 in log-loss at iteration 2), the first Adam steps amplify ANY difference by ~100x per step,
and the level of the median-of-last-21 PSNR moves by more than a dB between members of the family.  A per-slice offset of that
size between HIP and ONE member is therefore not evidence of an error (tests/test_gpu_ops.py::test_cells_vs_device_oracle_draws
carries those cells as expected failures; the mean over slices is asserted).  CPU only: these tests read the fixture."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import immoco_oracle as orc  # noqa: E402

FIX = os.path.join(ROOT, "tests", "golden", "c2_device_oracle_draws.npz")


def _stat(psnr):
    return np.median(psnr[:, 179:200].astype(np.float64), axis=1)


def _mean_se(x):
    return float(np.mean(x)), float(np.std(x, ddof=1) / np.sqrt(len(x)))


@pytest.mark.parametrize("dims,net", [(2, "image"), (3, "motion")])
def test_splitk_member_equals_the_plain_oracle_to_rounding(dims, net):
    """One forward / backward of both INRs: the split-K member differs from the plain oracle by fp32 rounding only."""
    cfg = orc.network_config if net == "image" else orc.mot_network_config
    torch.manual_seed(0)
    x = torch.rand(700, dims)
    a = orc.OracleINR(dims, 2, orc.encoding_config, cfg, backend="torch")
    w = torch.randn(700, 2)
    with torch.no_grad():                       # beyond the 1e-4 initialisation, so that the hidden layer is exercised
        a.params.mul_(50.0)
    ya = a(x)
    (ya * w).sum().backward()
    for c in (2, 4, 8):
        b = orc.OracleINR(dims, 2, orc.encoding_config, cfg, backend="torch", mlp_splitk=c)
        with torch.no_grad():
            b.params.copy_(a.params)
        yb = b(x)
        (yb * w).sum().backward()
        assert float((ya - yb).detach().abs().max()) <= 2e-6 * float(ya.detach().abs().max()), (net, c)
        gd = float((a.params.grad - b.params.grad).norm() / a.params.grad.norm())
        assert 0.0 < gd <= 1e-6, (net, c, gd)   # different rounding (not bit-identical), same sums


@pytest.mark.parametrize("sl", [2, 4])
def test_oracle_family_level_moves_with_the_summation_order(sl):
    g = np.load(FIX)
    keys = [f"s{sl}_it200_psnr_sk{c}" for c in (2, 4, 8)]
    if not all(k in g for k in keys):
        pytest.skip(f"no split-K ensembles for slice {sl} in the fixture")
    base = _stat(g[f"s{sl}_it200_psnr"])
    fam = [_stat(g[k]) for k in keys]
    mb, sb = _mean_se(base)
    print(f"slice {sl}: plain oracle {mb:.3f} +- {sb:.3f} ({len(base)} draws)")
    for c, f in zip((2, 4, 8), fam):
        m, s = _mean_se(f)
        print(f"  split-K {c}: {m:.3f} +- {s:.3f} ({len(f)} draws), minus plain {m - mb:+.3f} +- {np.hypot(s, sb):.3f}")
    pooled = np.concatenate(fam)
    mp, sp = _mean_se(pooled)
    shift, se = mp - mb, float(np.hypot(sp, sb))
    print(f"  pooled split-K {mp:.3f} +- {sp:.3f}; shift {shift:+.3f} +- {se:.3f} ({shift / se:+.1f} s.e.)")
    # the perturbation itself is tiny: identical start, the ensembles' mean log-loss agrees to 1e-4 through iteration 3
    # (by iteration 5 the members are 5e-4 ... 3e-3 apart - up to 15 standard errors - and stay so through iteration 40: on
    # slice 4 the members that are > 1.5e-3 above the plain oracle there end 1.5 dB below it, the one at 5e-4 ends level)
    lo = np.log(g[f"s{sl}_it200_loss"].astype(np.float64))
    for c in (2, 4, 8):
        lf = np.log(g[f"s{sl}_it200_loss_sk{c}"].astype(np.float64))
        assert abs(lf[:, 0].mean() - lo[:, 0].mean()) <= 1e-7
        assert np.abs(lf[:, :4].mean(axis=0) - lo[:, :4].mean(axis=0)).max() <= 1e-4, c
    # ... the spread of the plain oracle's draws at iteration 2 is the order of its atomics alone
    assert lo[:, 2].std(ddof=1) <= 1e-7
    # ... and the 200-iteration level is not common to the family (measured: slice 2 pooled -1.40 +- 0.44; slice 4 members
    # -1.48 +- 0.42, +0.26 +- 0.36, -1.48 +- 0.55, pooled -0.86 +- 0.33)
    assert abs(shift) >= 2.0 * se, (shift, se)
    means = [m for m, _ in map(_mean_se, [base] + fam)]
    assert max(means) - min(means) >= 1.0, means
