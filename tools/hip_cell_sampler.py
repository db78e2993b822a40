#!/usr/bin/env python
"""Per-run statistics of N HIP solves in the cells the device-oracle fixture holds (GPU box; product API only):

    python tools/hip_cell_sampler.py --cell plateau --slice 1 --runs 64 [--precision f32|f16mlp|bf16x2] --out gpurun_out/cells/x.npz
    python tools/hip_cell_sampler.py --cell it200 --slice 4 --runs 64 ...

cell "plateau": slice's 3000-iteration solve run to iteration 1000; per run the median PSNR over iterations 600, 625, ...,
975 and the loss of every iteration.  cell "it200": the reference's `iters=200` (src/test/test_immoco.py:65-72); per run
the median PSNR over the last 21 iterations, the final PSNR and the loss."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import miccai24_immoco_amd as pkg
from miccai24_immoco_amd import synth
from miccai24_immoco_amd.models.immoco import get_solver
from miccai24_immoco_amd.utils.motion_utils import masks_to_col_group
from miccai24_immoco_amd.utils.sampling import hip_psnr_samples, summarize


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cell", choices=["plateau", "it200"], required=True)
    ap.add_argument("--slice", type=int, required=True)
    ap.add_argument("--runs", type=int, default=64)
    ap.add_argument("--precision", choices=["f32", "f16mlp", "bf16x2"], default="f32")
    ap.add_argument("--seed", type=int, default=1337)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    if a.slice == 1:
        g = np.load(os.path.join(ROOT, "tests", "golden", "c2_slice1_input.npz"))
        k, lines = torch.from_numpy(g["kspace"]).to(dev), torch.from_numpy(g["lines"]).to(dev)
    else:
        from oracle import synth_cpu          # the device-oracle draws' own input (CPU generator), bit for bit
        s = synth_cpu.make_slice(320, 320, 10, a.slice)
        k, lines = s["kspace"].to(dev), s["lines"].to(dev)
    masks = pkg.extract_movement_groups(lines, make_list=True)
    gt = synth.phantom(320, 320, 1000 + a.slice).abs()
    sol = get_solver(dev, 320, 320, int(masks.shape[0]), mlp_fp16={"f32": 0, "f16mlp": 1, "bf16x2": 2}[a.precision])
    kin, cg = k / k.abs().max() * 16000, masks_to_col_group(masks)
    stat, fin, losses = [], [], []
    for r in range(a.runs):
        if a.cell == "plateau":
            grid = list(range(600, 1000, 25))
            ps, loss = hip_psnr_samples(sol, kin, cg, gt, 3000, grid + [999], seed=a.seed)
            stat.append(float(np.median([ps[t] for t in grid])))
            fin.append(ps[999])
        else:
            smp = list(range(179, 200))
            ps, loss = hip_psnr_samples(sol, kin, cg, gt, 200, smp, seed=a.seed)
            stat.append(float(np.median([ps[t] for t in smp])))
            fin.append(ps[199])
        losses.append(loss.astype(np.float32))
    print(f"HIP {a.precision} cell {a.cell} slice {a.slice} seed {a.seed}: {a.runs} runs, statistic mean %.3f sd %.3f se %.3f" % summarize(stat),
          "low (< 38 dB) %d" % sum(v < 38 for v in stat) if a.cell == "plateau" else "", flush=True)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        np.savez_compressed(a.out, stat=np.array(stat, dtype=np.float32), final=np.array(fin, dtype=np.float32),
                            loss=np.array(losses), cell=a.cell, slice_idx=np.int32(a.slice), precision=a.precision,
                            init_seed=np.int32(a.seed))


if __name__ == "__main__":
    main()
