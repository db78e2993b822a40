#!/usr/bin/env python
"""tests/golden/c2_oracle_slice1_3000it.npz from the CPU-oracle run (tools/oracle_c2.py 1 3000 out.npz) and,
for the record, the HIP runs of tools/diag_traj3000.py: loss of EVERY oracle iteration (float32), PSNR every 25.
    python tools/make_c2_3000_fixture.py /tmp/oracle_c2_s1_3000.npz gpurun_out/traj3000_hip_slice1.npz"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
o = np.load(sys.argv[1])
h = np.load(sys.argv[2])
assert int(o["iters_done"]) == 3000, int(o["iters_done"])
out = {"oracle_loss": o["loss"].astype(np.float32), "oracle_psnr_iters": o["psnr_iters"].astype(np.int32),
       "oracle_psnr": o["psnr"].astype(np.float32), "slice_idx": np.int32(o["slice_idx"]),
       "hip_iters": h["iters"].astype(np.int32), "hip_loss": h["loss"].astype(np.float32),
       "hip_psnr": h["psnr"].astype(np.float32)}
p = os.path.join(ROOT, "tests", "golden", "c2_oracle_slice1_3000it.npz")
np.savez_compressed(p, **out)
print(p, os.path.getsize(p), "bytes; oracle final loss", float(o["loss"][-1]), "psnr", float(o["psnr"][-1]))
