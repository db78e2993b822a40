"""Run the CPU oracle on config C2 (320x320, 10 groups) for N iterations; log loss/PSNR; save npz.

    python tools/oracle_c2.py <slice_idx> <iters> <out.npz> [order] [threads] [sched_iters] [redraw_seed] [init_seed]

`redraw_seed` >= 0: draw NEW summation orders before EVERY step (OracleIMMoCo.redraw: hash-grid backward block
order, MLP batch row order, motion-group order) from numpy's default_rng(redraw_seed) - the per-step analogue of
what nondeterministic atomics do; -1 (default) keeps the single fixed `order` for the whole trajectory.

`init_seed` (default 1337, the reference's fixed tcnn seed): seed of the initial parameters of both networks
(miccai24_immoco_amd's init_params(seed, seed) draws the same values).

`sched_iters` (default: iters) is the length of the solve whose lambda_GE schedule is used: `401 ... 3000` records
the first 401 iterations of a 3000-iteration solve (a draw of the metric's trajectory), not a 401-iteration solve.

`order` selects the fp32 summation order of the oracle's hash-grid backward (oracle/hashgrid_oracle.c:
0 ascending, 1 descending, k >= 2 strided blocks) and `threads` torch's thread count (changes the blocking of
the dense kernels): every (order, threads) pair is an equally valid fp32 evaluation of the same algorithm, i.e.
one DRAW of the chaotic trajectory.  The record also stores the slice's INPUT (corrupted k-space, voted lines),
so consumers (tests, bench.py's PSNR-delta report) do not need the oracle to regenerate it.
Build container only (about 2 s per iteration on 4 threads)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, torch.nn.functional as F
from oracle import immoco_oracle as orc, synth_cpu
idx, iters, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
order = int(sys.argv[4]) if len(sys.argv) > 4 else 0
threads = int(sys.argv[5]) if len(sys.argv) > 5 else int(os.environ.get("ORACLE_THREADS", "4"))
sched_iters = int(sys.argv[6]) if len(sys.argv) > 6 else iters
redraw_seed = int(sys.argv[7]) if len(sys.argv) > 7 else -1
init_seed = int(sys.argv[8]) if len(sys.argv) > 8 else 1337
rng = np.random.default_rng(redraw_seed) if redraw_seed >= 0 else None
torch.set_num_threads(threads)
s = synth_cpu.make_slice(320, 320, 10, idx)
masks = orc.extract_movement_groups(s["lines"], make_list=True)
gt = s["gt"].abs()
model = orc.OracleIMMoCo(masks,
                         image_inr=orc.OracleINR(2, 2, orc.encoding_config, orc.network_config, seed=init_seed, bwd_order=order),
                         motion_inr=orc.OracleINR(3, 2, orc.encoding_config, orc.mot_network_config, seed=init_seed, bwd_order=order))
k = s["kspace"]
kin = k.div(k.abs().max()).mul(16000).clone()
opt = torch.optim.Adam([{"params": model.motion_inr.parameters(), "lr": 1e-2}, {"params": model.image_inr.parameters(), "lr": 1e-2}])
lam = orc.lambda_schedule(sched_iters, 1e-2)
hist, psnrs, psnr_all = [], {}, []
t0 = time.time()
for j in range(iters):
    if rng is not None and j > 0:      # (the lattice plans exist after the first forward)
        model.redraw(rng)
    opt.zero_grad()
    kf, ip = model()
    loss = F.mse_loss(torch.view_as_real(kf), torch.view_as_real(kin)) + orc.gradient_entropy_loss(ip) * lam[j]
    loss.backward()
    opt.step()
    hist.append(float(loss.detach()))
    psnr_all.append(orc.crop_psnr(ip.detach().abs(), gt))
    if j % 25 == 0 or j == iters - 1:
        psnrs[j] = psnr_all[-1]
        print(j, f"loss {hist[-1]:.4f} psnr {psnrs[j]:.3f} t={time.time()-t0:.0f}s", flush=True)
    if j % 100 == 0 or j == iters - 1:
        np.savez_compressed(out, image=ip.detach().numpy(), kfm=kf.detach().numpy(), loss=np.array(hist, dtype=np.float64),
                            psnr_iters=np.array(list(psnrs.keys())), psnr=np.array(list(psnrs.values())), slice_idx=idx,
                            iters=iters, sched_iters=sched_iters, iters_done=j + 1, order=order, threads=threads, redraw_seed=redraw_seed, init_seed=init_seed,
                            psnr_all=np.array(psnr_all, dtype=np.float32),
                            kspace=k.numpy(), lines=s["lines"].numpy(), n_groups=int(masks.shape[0]))
