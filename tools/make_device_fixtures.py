#!/usr/bin/env python
"""tests/golden/c2_device_oracle_draws.npz from the records of tools/device_oracle_sampler.py (DEVICE-ORACLE draws: the
oracle's restatement evaluated by ATen on the GPU, fp32, nondeterministic atomics; every draw from the reference's one
initialisation, seed 1337):

    python tools/make_device_fixtures.py gpurun_out/dorc [gpurun_out/dorc2 ...]

  s{1,4,9,2,6,7}_it200_psnr / _loss  [draws, 200]  the reference script's iters=200 (src/test/test_immoco.py:65-72), every
                                               iteration (1 / 4 / 9: the pre-registered cells, 64 draws; 2 / 6 / 7: added after the
                                               first comparison to average the per-slice offsets over more slices, 48 draws)
  s1_plateau_psnr               [draws, 201]   slice 1, the metric's 3000-iteration solve, iterations 0, 5, ..., 1000
  s1_plateau_loss               [draws, 1001]  every iteration
  *_f16 variants                               the same cells with OracleINR(mlp_fp16=True) where drawn
  s{2,4}_it200_psnr_sk{2,4,8} / _loss_sk*  [32, 200]  the oracle with mlp_splitk = 2 / 4 / 8 (tests/test_oracle_family.py)
"""
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
srcs = sys.argv[1:]
out = {}


def merge(pattern):
    recs = [np.load(f) for src in srcs for f in sorted(glob.glob(os.path.join(src, pattern)))]
    recs = [r for r in recs if r["psnr"].shape[0] > 0]
    if not recs:
        return None
    r0 = recs[0]
    for r in recs:
        assert int(r["slice_idx"]) == int(r0["slice_idx"]) and int(r["sched_iters"]) == int(r0["sched_iters"])
        assert int(r["init_seed"]) == 1337 and int(r["mlp_fp16"]) == int(r0["mlp_fp16"])
        assert abs(float(r["kspace_abs_sum"]) - float(r0["kspace_abs_sum"])) <= 1e-9 * float(r0["kspace_abs_sum"])
    return dict(psnr=np.concatenate([r["psnr"] for r in recs]), loss=np.concatenate([r["loss"] for r in recs]),
                kspace_abs_sum=float(r0["kspace_abs_sum"]), n_groups=int(r0["n_groups"]))


for sl in (1, 4, 9, 2, 6, 7):
    for tag, suffix in (("f32", ""), ("f16", "_f16")):
        m = merge(f"s{sl}_200_{tag}*.npz")
        if m is None:
            continue
        assert m["psnr"].shape[1] == 200
        out[f"s{sl}_it200_psnr{suffix}"] = m["psnr"].astype(np.float32)
        out[f"s{sl}_it200_loss{suffix}"] = m["loss"].astype(np.float32)
        out[f"s{sl}_kspace_abs_sum"] = np.float64(m["kspace_abs_sum"])
        out[f"s{sl}_n_groups"] = np.int32(m["n_groups"])
        st = np.median(m["psnr"][:, 179:200], axis=1)
        print(f"slice {sl} it200 {tag}: {len(st)} draws, median-of-last-21 PSNR mean {st.mean():.3f} sd {st.std(ddof=1):.3f} "
              f"se {st.std(ddof=1) / np.sqrt(len(st)):.3f}")
# the oracle FAMILY: the same restatement with its MLP products summed over c interleaved slices of the inner dimension
# (OracleINR(mlp_splitk=c): another equally valid fp32 evaluation order), 32 draws each
for sl in (2, 4):
    for c in (2, 4, 8):
        m = merge(f"s{sl}_200_sk{c}.npz")
        if m is None:
            continue
        out[f"s{sl}_it200_psnr_sk{c}"] = m["psnr"].astype(np.float32)
        out[f"s{sl}_it200_loss_sk{c}"] = m["loss"].astype(np.float32)
        st = np.median(m["psnr"][:, 179:200], axis=1)
        print(f"slice {sl} it200 split-K {c}: {len(st)} draws, median-of-last-21 PSNR mean {st.mean():.3f} sd {st.std(ddof=1):.3f} "
              f"se {st.std(ddof=1) / np.sqrt(len(st)):.3f}")
for tag, suffix in (("f32", ""), ("f16", "_f16")):
    m = merge(f"s1_plateau_{tag}*.npz")
    if m is None:
        continue
    assert m["psnr"].shape[1] == 1001
    out[f"s1_plateau_psnr{suffix}"] = m["psnr"][:, ::5].astype(np.float32)
    out[f"s1_plateau_loss{suffix}"] = m["loss"].astype(np.float32)
    st = np.median(m["psnr"][:, 600:1000:25], axis=1)
    print(f"slice 1 plateau {tag}: {len(st)} draws, median PSNR over 600..975 mean {st.mean():.3f} sd {st.std(ddof=1):.3f} "
          f"se {st.std(ddof=1) / np.sqrt(len(st)):.3f}; low (< 38 dB): {int((st < 38).sum())} of {len(st)}")
out["psnr_plateau_iters"] = np.arange(0, 1001, 5, dtype=np.int32)
dst = os.path.join(ROOT, "tests", "golden", "c2_device_oracle_draws.npz")
np.savez_compressed(dst, **out)
print(dst, os.path.getsize(dst), "bytes", sorted(out))
