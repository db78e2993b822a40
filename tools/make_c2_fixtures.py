#!/usr/bin/env python
"""tests/golden/c2_*.npz from CPU-oracle records made by tools/oracle_c2.py (slice 1, config C2):

    python tools/make_c2_fixtures.py /tmp/orc

  c2_slice1_input.npz          the slice's INPUT (corrupted k-space c64, voted lines) - what every record was run on
  c2_oracle_slice1_draws.npz   >= 6 draws of the first 401 iterations of the 3000-iteration solve (loss of every
                               iteration, PSNR every 25): different fp32 summation orders / thread counts
  c2_oracle_slice1_3000it.npz  every full 3000-iteration record found (loss of every iteration, PSNR every 25)
"""
import glob, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
src = sys.argv[1]
draws = [np.load(f) for f in sorted(glob.glob(os.path.join(src, "s1_d401_*.npz")))]
full = [np.load(f) for f in sorted(glob.glob(os.path.join(src, "s1_3000_*.npz")))]
draws = [d for d in draws if int(d["iters_done"]) == 401 and int(d["sched_iters"]) == 3000]
full = [d for d in full if int(d["iters_done"]) == 3000]
assert draws and full, (len(draws), len(full))
ref = full[0]
for d in draws + full:
    assert int(d["slice_idx"]) == 1 and np.array_equal(d["kspace"], ref["kspace"]) and np.array_equal(d["lines"], ref["lines"])
np.savez_compressed(os.path.join(OUT, "c2_slice1_input.npz"), kspace=ref["kspace"].astype(np.complex64),
                    lines=ref["lines"], slice_idx=np.int32(1), n_groups=np.int32(ref["n_groups"]))
# the first 401 iterations of every full record are draws too (unless a draw with the same summation order and
# thread count exists: the oracle is deterministic, that would be the same trajectory twice)
seen = {(int(d["order"]), int(d["threads"])) for d in draws}
heads = [d for d in full if (int(d["order"]), int(d["threads"])) not in seen]
loss = [d["loss"][:401] for d in draws] + [d["loss"][:401] for d in heads]
psnr = [d["psnr"][:17] for d in draws] + [d["psnr"][:17] for d in heads]
assert all(np.array_equal(d["psnr_iters"][:17], np.arange(0, 401, 25)) for d in draws + full)
np.savez_compressed(os.path.join(OUT, "c2_oracle_slice1_draws.npz"), loss=np.array(loss, dtype=np.float32),
                    psnr_iters=np.arange(0, 401, 25, dtype=np.int32), psnr=np.array(psnr, dtype=np.float32),
                    order=np.array([int(d["order"]) for d in draws + heads], dtype=np.int32),
                    threads=np.array([int(d["threads"]) for d in draws + heads], dtype=np.int32), slice_idx=np.int32(1))
np.savez_compressed(os.path.join(OUT, "c2_oracle_slice1_3000it.npz"),
                    oracle_loss=np.array([d["loss"] for d in full], dtype=np.float32),
                    oracle_psnr_iters=full[0]["psnr_iters"].astype(np.int32),
                    oracle_psnr=np.array([d["psnr"] for d in full], dtype=np.float32),
                    order=np.array([int(d["order"]) for d in full], dtype=np.int32),
                    threads=np.array([int(d["threads"]) for d in full], dtype=np.int32), slice_idx=np.int32(1))
for f in ("c2_slice1_input", "c2_oracle_slice1_draws", "c2_oracle_slice1_3000it"):
    print(f, os.path.getsize(os.path.join(OUT, f + ".npz")), "bytes")
L = np.array(loss)
for j in (0, 5, 25, 50, 100, 200, 400):
    print(f"it {j:4d}: loss min {L[:, j].min():.4f} max {L[:, j].max():.4f}")
print("psnr @400:", np.array(psnr)[:, 16], " end-of-solve psnr:", [float(d["psnr"][-1]) for d in full])
