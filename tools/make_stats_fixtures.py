#!/usr/bin/env python
"""tests/golden fixtures of the statistical parity tests from CPU-oracle runs made by tools/oracle_c2.py:

    python tools/make_stats_fixtures.py /tmp/orc3 [/tmp/orc4]

  c2_oracle_200it_draws.npz        the reference script's own setting (iters = 200, /root/reference/src/test/
                                   test_immoco.py:65-72) at 320x320 / 10 groups: per slice (1, 4, 9) the loss and the
                                   PSNR of EVERY iteration of 8 draws (fp32 summation orders 0, 1, 2, 3, 5, 7, 11, 13 of
                                   the hash-grid backward), plus a checksum of the input the draws were run on; for slice 1
                                   also 8 draws with summation orders re-drawn before every step (`*_redraw`); for
                                   slices 4 and 9 (second directory) re-drawn-order draws from init seeds 2001...2008
                                   (`s4_*_initseed`, `s9_*_initseed`)
  c2_oracle_slice1_redraw1400.npz  the first 1400 iterations of the 3000-iteration solve of slice 1 with NEW summation
                                   orders drawn before EVERY step (OracleIMMoCo.redraw): 6 draws, loss and PSNR of every
                                   iteration
  c2_oracle_slice1_initseeds.npz   (second directory) the first 1001 iterations of the same solve from OTHER initial
                                   parameters (init seeds 2001...2008 instead of the reference's fixed 1337; orders
                                   re-drawn every step): init_seed, redraw_seed, loss and PSNR of every iteration
"""
import glob, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
src = sys.argv[1]
out = {}
for sl in (1, 4, 9):
    fs = sorted(glob.glob(os.path.join(src, f"s{sl}_200_o*.npz")))
    ds = [np.load(f) for f in fs]
    ds = [d for d in ds if int(d["iters_done"]) == 200 and int(d["sched_iters"]) == 200 and int(d["redraw_seed"]) < 0]
    if not ds:
        continue
    for d in ds:
        assert int(d["slice_idx"]) == sl and np.array_equal(d["kspace"], ds[0]["kspace"]) and np.array_equal(d["lines"], ds[0]["lines"])
    out[f"s{sl}_loss"] = np.array([d["loss"] for d in ds], dtype=np.float32)
    out[f"s{sl}_psnr"] = np.array([d["psnr_all"] for d in ds], dtype=np.float32)
    out[f"s{sl}_order"] = np.array([int(d["order"]) for d in ds], dtype=np.int32)
    out[f"s{sl}_kspace_abs_sum"] = np.float64(np.abs(ds[0]["kspace"]).astype(np.float64).sum())
    out[f"s{sl}_n_groups"] = np.int32(ds[0]["n_groups"])
    rs = [np.load(f) for f in sorted(glob.glob(os.path.join(src, f"s{sl}_200rd_*.npz")))]
    rs = [d for d in rs if int(d["iters_done"]) == 200 and int(d["sched_iters"]) == 200 and int(d["redraw_seed"]) >= 0
          and np.array_equal(d["kspace"], ds[0]["kspace"])]
    if rs:   # the same with summation orders re-drawn before every step (OracleIMMoCo.redraw)
        out[f"s{sl}_loss_redraw"] = np.array([d["loss"] for d in rs], dtype=np.float32)
        out[f"s{sl}_psnr_redraw"] = np.array([d["psnr_all"] for d in rs], dtype=np.float32)
        pr = out[f"s{sl}_psnr_redraw"]
        print(f"slice {sl}: {len(rs)} redraw draws, final PSNR {np.round(pr[:, -1], 2)} mean {pr[:, -1].mean():.3f}; "
              f"median of last 21: {np.round(np.median(pr[:, 179:], axis=1), 2)} mean {np.median(pr[:, 179:], axis=1).mean():.3f}")
    p = out[f"s{sl}_psnr"]
    print(f"slice {sl}: {len(ds)} draws, final PSNR {np.round(p[:, -1], 2)} mean {p[:, -1].mean():.3f} sd {p[:, -1].std(ddof=1):.3f}; "
          f"median of last 21: {np.round(np.median(p[:, 179:], axis=1), 2)}")
for sl in ((4, 9) if len(sys.argv) > 2 else ()):
    # the 200-iteration schedule from OTHER initial parameters (init seeds 2001...2008, re-drawn orders)
    rs = [np.load(f) for f in sorted(glob.glob(os.path.join(sys.argv[2], f"s{sl}_200_init*.npz")))]
    rs = [d for d in rs if int(d["iters_done"]) == 200 and int(d["sched_iters"]) == 200 and int(d["redraw_seed"]) >= 0]
    if rs and f"s{sl}_loss" in out:
        ref = np.load(sorted(glob.glob(os.path.join(src, f"s{sl}_200_o*.npz")))[0])
        assert all(np.array_equal(d["kspace"], ref["kspace"]) for d in rs)
        out[f"s{sl}_initseed"] = np.array([int(d["init_seed"]) for d in rs], dtype=np.int32)
        out[f"s{sl}_psnr_initseed"] = np.array([d["psnr_all"] for d in rs], dtype=np.float32)
        out[f"s{sl}_loss_initseed"] = np.array([d["loss"] for d in rs], dtype=np.float32)
        sd_, m = out[f"s{sl}_initseed"], np.median(out[f"s{sl}_psnr_initseed"][:, 179:], axis=1)
        for sd in sorted(set(sd_.tolist())):
            print(f"slice {sl}, init seed {sd}: median of last 21 {np.round(m[sd_ == sd], 2).tolist()}")
        print(f"slice {sl}: mean over seeds of the per-seed means %.3f" % np.mean([m[sd_ == sd].mean() for sd in set(sd_.tolist())]))
if out:
    np.savez_compressed(os.path.join(OUT, "c2_oracle_200it_draws.npz"), **out)
    print("c2_oracle_200it_draws.npz", os.path.getsize(os.path.join(OUT, "c2_oracle_200it_draws.npz")), "bytes")
fs = sorted(glob.glob(os.path.join(src, "s1_rd1400_*.npz")))
ds = [np.load(f) for f in fs]
ds = [d for d in ds if int(d["iters_done"]) == 1400 and int(d["sched_iters"]) == 3000 and int(d["redraw_seed"]) >= 0]
if ds:
    np.savez_compressed(os.path.join(OUT, "c2_oracle_slice1_redraw1400.npz"),
                        loss=np.array([d["loss"] for d in ds], dtype=np.float32),
                        psnr=np.array([d["psnr_all"] for d in ds], dtype=np.float32),
                        redraw_seed=np.array([int(d["redraw_seed"]) for d in ds], dtype=np.int32), slice_idx=np.int32(1))
    p = np.array([d["psnr_all"] for d in ds])
    print(f"redraw: {len(ds)} draws; PSNR@1399 {np.round(p[:, -1], 2)}; median(1350..1399) {np.round(np.median(p[:, 1350:], axis=1), 2)}; "
          f"samples < 38 dB between 100 and 1399: {(p[:, 100:] < 38).mean():.3f}")
    rec = np.load(os.path.join(OUT, "c2_oracle_slice1_3000it.npz"))
    it = list(rec["oracle_psnr_iters"])
    o = rec["oracle_psnr"]
    print("fixed-order records: PSNR@1400", np.round(o[:, it.index(1400)], 2), "every-25 samples < 38 dB between 100 and 1500:",
          float((o[:, it.index(100):it.index(1500) + 1] < 38).mean()))
    print("redraw every-25 samples < 38 dB between 100 and 1375:", float((p[:, 100:1400:25] < 38).mean()),
          " sd of PSNR@1375 across draws: redraw %.3f fixed %.3f" % (p[:, 1375].std(ddof=1), o[:, it.index(1375)].std(ddof=1)))

if len(sys.argv) > 2:
    fs = sorted(glob.glob(os.path.join(sys.argv[2], "s1_init*.npz")))
    ds = [np.load(f) for f in fs]
    ds = [d for d in ds if int(d["iters_done"]) >= 1001 and int(d["sched_iters"]) == 3000 and int(d["redraw_seed"]) >= 0]
    if ds:
        seeds = np.array([int(d["init_seed"]) for d in ds], dtype=np.int32)
        p = np.array([d["psnr_all"][:1001] for d in ds], dtype=np.float32)
        np.savez_compressed(os.path.join(OUT, "c2_oracle_slice1_initseeds.npz"), init_seed=seeds,
                            redraw_seed=np.array([int(d["redraw_seed"]) for d in ds], dtype=np.int32),
                            loss=np.array([d["loss"][:1001] for d in ds], dtype=np.float32), psnr=p, slice_idx=np.int32(1))
        plat = np.median(p[:, 600:1000:25], axis=1)
        for sd in sorted(set(seeds.tolist())):
            print(f"init seed {sd}: plateau medians (600..975 every 25) {np.round(plat[seeds == sd], 2).tolist()}")
