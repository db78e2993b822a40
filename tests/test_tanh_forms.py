"""The two tanh forms of csrc/mlp_mfma.hip, emulated in numpy float32 (CPU): the shipped ten-instruction form
(x - x^3/3 below 0.04, 1 - 2/(exp(2|x|)+1) above) and the polynomial variant behind -DIMMOCO_DIAG_POLY_TANH.  The constants
are read from the source, so the accuracy figures quoted there and in DESIGN.md stay attached to what is compiled.  (numpy's
exp2 and division are correctly rounded where v_exp_f32 / v_rcp_f32 are 1-ulp instructions: the emulation is a lower bound of
the exp form's error and exact for the polynomial, which is fused multiply-adds only.)"""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = open(os.path.join(ROOT, "miccai24_immoco_amd", "csrc", "mlp_mfma.hip")).read()
f32 = np.float32


def _tanh_fast_source():
    body = SRC[SRC.index("__device__ __forceinline__ float tanh_fast(float x)"):]
    return body[:body.index("\n}\n")]


def _fma(a, b, c):     # one rounding, like v_fma_f32
    return (a.astype(np.float64) * b.astype(np.float64) + np.asarray(c, dtype=np.float64)).astype(f32)


def _big(ax):
    e = np.exp2((ax * f32(2.885390082)).astype(f32)).astype(f32)
    return _fma(f32(-2.0) * np.ones_like(ax), (f32(1.0) / (e + f32(1.0))).astype(f32), f32(1.0))


def _shipped(ax):
    body = _tanh_fast_source()
    m = re.search(r"#else\s+const float small = ax \* fmaf\(ax \* ax, (-?[0-9.]+)f, 1\.f\);\s+return copysignf\(ax < ([0-9.]+)f \? small : big, x\);", body)
    assert m, "shipped tanh form not found in csrc/mlp_mfma.hip"
    c, cut = f32(m.group(1)), f32(m.group(2))
    small = (ax * _fma((ax * ax).astype(f32), c * np.ones_like(ax), f32(1.0))).astype(f32)
    return np.where(ax < cut, small, _big(ax)), float(cut)


def _poly(ax):
    body = _tanh_fast_source()
    m = re.search(r"const float p = fmaf\(fmaf\(fmaf\((-?[0-9.]+)f, u, (-?[0-9.]+)f\), u, (-?[0-9.]+)f\), u, (-?[0-9.]+)f\);\s+"
                  r"const float small = fmaf\(ax \* u, p, ax\);\s+return copysignf\(ax < ([0-9.]+)f \? small : big, x\);", body)
    assert m, "polynomial tanh form not found in csrc/mlp_mfma.hip"
    c3, c2, c1, c0, cut = (f32(m.group(i)) for i in range(1, 6))
    u = (ax * ax).astype(f32)
    one = np.ones_like(ax)
    p = _fma(_fma(_fma(c3 * one, u, c2), u, c1), u, c0)
    small = _fma((ax * u).astype(f32), p, ax)
    return np.where(ax < cut, small, _big(ax)), float(cut)


def _rel(y, x):
    ref = np.tanh(x.astype(np.float64))
    return np.abs(y.astype(np.float64) - ref) / ref


def test_tanh_forms_accuracy():
    rng = np.random.default_rng(0)
    x = np.exp(rng.uniform(np.log(1e-4), np.log(6.0), 1_000_000)).astype(f32)
    ys, cut_s = _shipped(x)
    yp, cut_p = _poly(x)
    assert cut_s == np.float32(0.04) and cut_p == 0.5
    rs, rp = _rel(ys, x), _rel(yp, x)
    rows = []
    for lo, hi in ((1e-4, 0.04), (0.04, 0.1), (0.1, 0.3), (0.3, 0.5), (0.5, 6.0)):
        m = (x >= lo) & (x < hi)
        rows.append((lo, hi, float(np.sqrt((rs[m] ** 2).mean())), float(rs[m].max()), float(np.sqrt((rp[m] ** 2).mean())), float(rp[m].max())))
        print("[%g, %g): shipped rms %.1e max %.1e | polynomial rms %.1e max %.1e" % rows[-1])
    # the shipped form: an ulp below 0.04, the cancellation of 1 - 2/(e+1) above it (what DESIGN.md 2.5 quotes)
    assert rows[0][3] <= 5e-7
    assert 4e-7 <= rows[1][2] <= 1.5e-6 and rows[1][3] <= 5e-6          # [0.04, 0.1): rms 8e-7
    assert 1e-7 <= rows[2][2] <= 5e-7                                    # [0.1, 0.3): rms 2.6e-7
    # the polynomial: the rounding floor everywhere below its cut
    for r in rows[:4]:
        assert r[4] <= 5e-8 and r[5] <= 1.5e-7, r
    # both use the same exp form above 0.5
    m = x >= 0.5
    assert np.array_equal(ys[m], yp[m])
