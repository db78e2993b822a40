#!/usr/bin/env python
"""Draws of the DEVICE ORACLE (oracle/immoco_oracle.py with device="cuda": ATen kernels only, nondeterministic
fp32 atomics in the hash-grid backward - no kernel of libimmoco_hip.so) for the statistical parity fixtures.
TEST INFRASTRUCTURE: runs on the GPU box, writes per-draw records under gpurun_out/; tools/make_device_fixtures.py
turns them into tests/golden/c2_device_oracle_draws.npz.

    python tools/device_oracle_sampler.py --slice 1 --sched 3000 --run 1001 --draws 16 --out gpurun_out/dorc/s1_plateau_f32
    python tools/device_oracle_sampler.py --slice 4 --sched 200 --draws 32 --out gpurun_out/dorc/s4_200_f32 [--mlp-fp16]

Every draw starts from the reference's ONE initialisation (tiny-cuda-nn's module default seed 1337 for both INRs,
/root/reference/src/models/immoco.py:60-65) unless --seed says otherwise; what differs between draws is the order in
which the atomics of `index_add_` land.  Records: loss and crop-PSNR (src/test/test_immoco.py:74-85) of EVERY iteration's
forward.  The file is rewritten after every draw, so a killed call keeps what it had."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.nn.functional as F

from oracle import immoco_oracle as orc, synth_cpu
from miccai24_immoco_amd.synth import phantom


def slice_input(sl):
    """(kspace c64 [320,320], voted lines bool [320], ground-truth magnitude) of C2 slice `sl` - slice 1 from the
    committed input of the CPU-oracle records, the others regenerated on the CPU (same seeds as the tests)."""
    if sl == 1:
        g = np.load(os.path.join(ROOT, "tests", "golden", "c2_slice1_input.npz"))
        k, lines = torch.from_numpy(g["kspace"]), torch.from_numpy(g["lines"])
    else:
        s = synth_cpu.make_slice(320, 320, 10, sl)
        k, lines = s["kspace"], s["lines"]
    return k, lines, phantom(320, 320, 1000 + sl).abs()


def device_psnr(pred_abs, gt_abs):
    """crop_psnr (oracle: test_immoco.py:74-85, evaluate.py:19-47) as device tensors, no synchronisation."""
    H, W = gt_abs.shape
    c0, c1 = H // 4, W // 4
    p, g = pred_abs[c0:-c0, c1:-c1], gt_abs[c0:-c0, c1:-c1]
    p = (p - p.min()) / (p.max() - p.min() + 1e-24)
    g = (g - g.min()) / (g.max() - g.min() + 1e-24)
    return 20.0 * torch.log10(1.0 / torch.sqrt(torch.mean((p - g) ** 2)))


_MODELS = {}


TANH = "torch"
MLP_F64 = False
MLP_SPLITK = 0


def fresh_model(masks, seed, mlp_fp16, dev):
    """One model per (shape, seed, precision) and process: the coordinate plans (numpy hashing of 1 M points, seconds)
    are built once, every draw restarts from the initial parameters."""
    key = (tuple(masks.shape), seed, bool(mlp_fp16))
    if key not in _MODELS:
        kw = dict(seed=seed, device=dev, mlp_fp16=mlp_fp16, tanh=TANH, mlp_f64=MLP_F64, mlp_splitk=MLP_SPLITK)
        m = orc.OracleIMMoCo(masks, image_inr=orc.OracleINR(2, 2, orc.encoding_config, orc.network_config, **kw),
                             motion_inr=orc.OracleINR(3, 2, orc.encoding_config, orc.mot_network_config, **kw))
        _MODELS[key] = (m, m.image_inr.params.detach().clone(), m.motion_inr.params.detach().clone())
    m, pi0, pm0 = _MODELS[key]
    with torch.no_grad():
        m.image_inr.params.copy_(pi0)
        m.motion_inr.params.copy_(pm0)
    m.image_inr.params.grad = m.motion_inr.params.grad = None
    return m


def one_draw(k, masks, gt, sched, run, seed, mlp_fp16, dev):
    model = fresh_model(masks, seed, mlp_fp16, dev)
    kin = (k / k.abs().max() * 16000).to(dev)
    lam = orc.lambda_schedule(sched, 1e-2)
    opt = torch.optim.Adam([{"params": model.motion_inr.parameters(), "lr": 1e-2},
                            {"params": model.image_inr.parameters(), "lr": 1e-2}])
    loss_rec = torch.zeros(run, device=dev)
    psnr_rec = torch.zeros(run, device=dev)
    gt_d = gt.to(dev)
    for j in range(run):
        opt.zero_grad()
        kf, ip = model()
        loss = F.mse_loss(torch.view_as_real(kf), torch.view_as_real(kin)) + orc.gradient_entropy_loss(ip) * lam[j]
        loss.backward()
        opt.step()
        with torch.no_grad():
            loss_rec[j] = loss.detach()
            psnr_rec[j] = device_psnr(ip.detach().abs(), gt_d)
    return loss_rec.cpu().numpy(), psnr_rec.cpu().numpy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slice", type=int, required=True)
    ap.add_argument("--sched", type=int, required=True, help="iterations of the lambda schedule (iters of the solve)")
    ap.add_argument("--run", type=int, default=0, help="iterations actually run (default: all of the schedule)")
    ap.add_argument("--draws", type=int, default=8)
    ap.add_argument("--seed", type=int, default=1337)
    ap.add_argument("--mlp-fp16", action="store_true")
    ap.add_argument("--mlp-f64", action="store_true", help="MLP matrix products accumulated in float64 (sensitivity experiment)")
    ap.add_argument("--mlp-splitk", type=int, default=0,
                    help="MLP products summed over c interleaved slices of the inner dimension (sensitivity experiment)")
    ap.add_argument("--tanh", choices=["torch", "alt"], default="torch",
                    help="alt: an equally valid fp32 tanh formula (sensitivity experiment, DESIGN.md 2.4)")
    ap.add_argument("--budget-s", type=float, default=1e9, help="start no new draw after this many seconds")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    global TANH, MLP_F64, MLP_SPLITK
    TANH, MLP_F64, MLP_SPLITK = a.tanh, a.mlp_f64, a.mlp_splitk
    run = a.run or a.sched
    dev = torch.device("cuda", 0)
    k, lines, gt = slice_input(a.slice)
    masks = orc.extract_movement_groups(lines, make_list=True)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    losses, psnrs, t0 = [], [], time.time()
    for d in range(a.draws):
        if time.time() - t0 > a.budget_s:
            break
        t1 = time.time()
        l, p = one_draw(k, masks, gt, a.sched, run, a.seed, a.mlp_fp16, dev)
        losses.append(l)
        psnrs.append(p)
        np.savez_compressed(a.out, loss=np.array(losses, dtype=np.float32), psnr=np.array(psnrs, dtype=np.float32),
                            slice_idx=np.int32(a.slice), sched_iters=np.int32(a.sched), iters_run=np.int32(run),
                            init_seed=np.int32(a.seed), mlp_fp16=np.int32(a.mlp_fp16), n_groups=np.int32(masks.shape[0]),
                            kspace_abs_sum=np.float64(k.abs().double().sum()))
        print(f"slice {a.slice} sched {a.sched} run {run} mlp_fp16 {int(a.mlp_fp16)} draw {d}: {(time.time() - t1) / run * 1e3:.1f} ms/iteration, "
              f"loss[0] {l[0]:.4f} loss[-1] {l[-1]:.4f} psnr[-1] {p[-1]:.2f} median(last 21) {np.median(p[-21:]):.2f}", flush=True)


if __name__ == "__main__":
    main()
