#!/usr/bin/env python
"""When do two ensembles of 200-iteration solves separate?  Mean log-loss per iteration of a sample (records of
tools/device_oracle_sampler.py or tools/hip_cell_sampler.py) minus that of the device-oracle draws of the same slice in
tests/golden/c2_device_oracle_draws.npz, with the z-score of the difference and both ensembles' own spread:

    python tools/early_displacement.py <slice> name=glob [name=glob ...]

An ensemble whose spread comes from the order of fp32 atomics alone (the device oracle: 3e-8 in log-loss at iteration 2)
does not cover the difference between two fixed evaluation orders of the same fp32 sums (1e-7 ... 1e-6 per step); the first
Adam steps amplify either by ~100x per step."""
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sl = int(sys.argv[1])
g = np.load(os.path.join(ROOT, "tests", "golden", "c2_device_oracle_draws.npz"))
o_loss = g[f"s{sl}_it200_loss"].astype(np.float64)
o_psnr = np.median(g[f"s{sl}_it200_psnr"][:, 179:200].astype(np.float64), axis=1)
o = np.log(o_loss)
T = [0, 1, 2, 3, 5, 10, 20, 30, 40, 60, 100, 150, 199]
print(f"slice {sl}: device oracle, {len(o)} draws; sd of log-loss at t = " + ", ".join(f"{t}: {o[:, t].std(ddof=1):.1e}" for t in T))
print(f"median-of-last-21 PSNR: device oracle {o_psnr.mean():.3f} +- {o_psnr.std(ddof=1) / np.sqrt(len(o_psnr)):.3f}")
for arg in sys.argv[2:]:
    name, pat = arg.split("=", 1)
    loss, stat = [], []
    for f in sorted(glob.glob(pat)):
        r = np.load(f)
        loss.append(r["loss"].astype(np.float64))
        stat.append(r["stat"].astype(np.float64) if "stat" in r else np.median(r["psnr"][:, 179:200].astype(np.float64), axis=1))
    if not loss:
        print(name, "no records")
        continue
    h, st = np.log(np.concatenate(loss)), np.concatenate(stat)
    cells = []
    for t in T:
        d = h[:, t].mean() - o[:, t].mean()
        se = np.sqrt(h[:, t].var(ddof=1) / len(h) + o[:, t].var(ddof=1) / len(o))
        cells.append(f"t{t} {d:+.1e} (z {d / max(se, 1e-12):+.0f})")
    print(f"{name}: {len(h)} runs, PSNR statistic {st.mean():.3f} +- {st.std(ddof=1) / np.sqrt(len(st)):.3f} | mean log-loss minus the oracle's: " + "  ".join(cells))
