#!/usr/bin/env python3
"""bench.py — IM-MoCo per-slice inner optimisation loop on MI355X.

Metric (BASELINE.json): slices/s at 320x320, 10 motion groups, 3000 Adam iterations
(config C2).  A "step" is one full slice solve (immoco.py:116-206: normalisation, INR
init, 3000 iterations, last-forward outputs); inputs are synthetic (seeded phantom +
the reference's motion simulator) and resident in HBM before the timed region.

  python bench.py --gpus N --steps K --warmup W [--iters 3000] [--no-cpu-baseline]
                  [--workload c2|c3|c5] [--batch 64 --lanes 2]

The first timed slice of the default workload is slice 1 of tests/golden/c2_slice1_input.npz - the slice the
CPU oracle's full 3000-iteration records were run on - so the line carries the metric's second half,
`psnr_delta_db` (HIP minus oracle, PSNR against the synthetic ground truth).
--workload c3: BASELINE config 3, a step = one batch of --batch slices of the C2 shape through
immoco_solver_solve_batch; --workload c5: 640x640, 20 groups, fp16 tables (both informational lines).

N > 1: one rank per GPU; every rank solves its own K slices (weak scaling, no data-path
collective); the only collective is the final gather of the images (outside the hot loop,
inside the timed region).  `python bench.py --gpus N` launches its own N ranks (a parent that
never touches the GPU starts N fresh child processes with RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* set, relays rank 0's line and fails if any child does); under an outer
`torch.distributed.run` (WORLD_SIZE already set) it is simply one of the ranks.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H = W = 320
N_MOVEMENTS = 10
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# Arithmetic of the default line: exact fp32.  The reference's INRs run in fp16 (tiny-cuda-nn `__half` networks, loss
# scale 128: /root/reference/src/models/immoco.py:11-25,60-65), so the two faster MLP arithmetics of this build -
# "f16mlp" (fp16 operands, fp32 accumulation: 1.02 ms per iteration) and "bf16x2" (two-term bf16 split: 1.15 ms) - are
# never narrower than the reference.  north_star states an fp32 tolerance and the PSNR statistics resolve +-0.3 ... 0.5 dB,
# not 0.1 dB (from the reference's one initialisation f16mlp reads up to -1.4 dB at 200 iterations; over eight
# initialisations no statistic of either mode differs from fp32 by more than 0.5 dB), so they are reported as
# `other_precision` lines of the same run and the headline stays fp32.
DEFAULT_PRECISION = "f32"


def algorithmic_bytes(solver, nM):
    """Algorithmic HBM bytes per launch of each kernel of the iteration:
    compulsory reads/writes of the tensors the kernel consumes/produces, tables counted once."""
    P = H * W
    NP = nM * P
    npi, npm = solver.n_params_image, solver.n_params_motion
    tab_i, tab_m = 4 * (npi - 256 * 40), 4 * (npm - 64 * 48)
    return {
        "image_encode_fwd": tab_i + 128 * P,
        "image_mlp_fwd": 128 * P + 8 * P,
        "image_to_fft_slot": 16 * P,
        "motion_encode_fwd": tab_m + 128 * NP,
        "motion_mlp_fwd": 128 * NP + 8 * NP,
        "motion_warp_fwd": 8 * NP + 8 * P + 16 * NP,
        "fft_fwd": 16 * (nM + 1) * P,
        "select_dc_seed": 8 * P * 3 + 8 * (nM + 1) * P,
        "fft_adjoint": 16 * (nM + 1) * P,
        "image_grad_init_ge": 24 * P,
        "motion_warp_bwd": 8 * NP * 3 + 16 * P,
        "motion_mlp_bwd": 128 * NP * 2 + 8 * NP,
        "motion_encode_bwd": 128 * NP + 2 * tab_m,
        "image_mlp_bwd": 128 * P * 2 + 8 * P,
        "image_encode_bwd": 128 * P + 2 * tab_i,
        "adam_motion": 28 * npm,
        "adam_image": 28 * npi,
        "tick": 4,
    }


def cpu_baseline(iters_sample=30):
    """The oracle (CPU restatement of the reference loop) on this box's host cores, config C2,
    a bounded sample of iterations; slices/s extrapolated to 3000 iterations."""
    from oracle import immoco_oracle as orc, synth_cpu
    # torch's CPU kernels stop scaling (and then regress) far below the 256 hardware threads of the
    # GPU box: measured 38.7 s/iteration with 256 threads vs 7 s with 4; use at most 32
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    s = synth_cpu.make_slice(H, W, N_MOVEMENTS, 0)
    masks = orc.extract_movement_groups(s["lines"], make_list=True)
    model = orc.OracleIMMoCo(masks)
    k = s["kspace"]
    kin = k.div(k.abs().max()).mul(16000)
    opt = torch.optim.Adam([{"params": model.motion_inr.parameters(), "lr": 1e-2},
                            {"params": model.image_inr.parameters(), "lr": 1e-2}])
    import torch.nn.functional as F

    def one():
        opt.zero_grad()
        kf, ip = model()
        loss = F.mse_loss(torch.view_as_real(kf), torch.view_as_real(kin)) + orc.gradient_entropy_loss(ip) * 1e-2
        loss.backward()
        opt.step()
    one()  # builds the coordinate plans (not timed: constant per shape)
    t0 = time.perf_counter()
    for _ in range(iters_sample):
        one()
    dt = (time.perf_counter() - t0) / iters_sample
    return {"value": 1.0 / (3000 * dt), "unit": "slices/s", "cores": cores, "kind": "port",
            "sample": f"{iters_sample} Adam iterations of config C2 (320x320, {masks.shape[0]} groups) with the torch-CPU "
                      f"oracle, {dt:.2f} s/iter, extrapolated x3000 iterations"}


def launch_ranks(n, argv):
    """Parent of `python bench.py --gpus N` without an outer launcher: N child processes, one per GPU, over
    127.0.0.1.  The parent makes no HIP / torch.cuda call and loads no library (a process that has initialised
    the GPU must not be the one that forks the ranks); children inherit stdout / stderr, so rank 0's JSON line is
    this command's.  Exit status: 0 iff every rank exits 0; the first failure terminates the other ranks (their
    exact PIDs)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc, alive = 0, list(procs)
    while alive:
        time.sleep(0.2)
        for pr in list(alive):
            code = pr.poll()
            if code is None:
                continue
            alive.remove(pr)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for other in alive:
                    other.terminate()
    return rc


def dry_run(args):
    """--dry-run: the rank plumbing of the N > 1 path on the CPU (`gloo`), no GPU, no HIP library: every rank
    contributes K fake images of its own, the same barrier / MAX-over-ranks timing and the single gather run, rank
    0 prints the line.  tests/test_bench_launcher.py drives it at N = 2."""
    import torch.distributed as dist
    from miccai24_immoco_amd.shard import gather_images
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    K = args.steps
    t0 = time.perf_counter()
    local = torch.stack([torch.full((4, 4), float(rank * K + j), dtype=torch.complex64) for j in range(K)])
    allimgs = gather_images(local, K * world, dst=0 if args.gather == "rank0" else None)
    dt = time.perf_counter() - t0
    seen = dist.get_world_size() if dist.is_initialized() else 1
    if dist.is_initialized():
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        ids = sorted({int(v.real) for v in allimgs[:, 0, 0]})
        print(json.dumps({"metric": "dry-run (rank plumbing only, no GPU work)", "value": K * world / max(dt, 1e-9),
                          "unit": "fake slices/s", "n_gpus": world, "n_ranks_seen": seen, "steps": K, "warmup": args.warmup,
                          "images_gathered": int(allimgs.shape[0]), "image_ids": ids, "scaling": "weak",
                          "self_launched": bool(os.environ.get("BENCH_SELF_LAUNCHED"))}), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--iters", type=int, default=3000, help="Adam iterations per slice (BASELINE: 3000)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt-precision", action="store_true", help="skip the informational line in the other arithmetic")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--grad-parts", type=int, default=0, help="transposed-index parts (0 = library default)")
    ap.add_argument("--table-fp16", action="store_true",
                    help="fp16 hash-grid features (BASELINE config 5 precision); default fp32 like config 2")
    ap.add_argument("--precision", choices=["f32", "f16mlp", "bf16x2"], default=DEFAULT_PRECISION,
                    help="f32: exact fp32 everywhere; f16mlp: both MLPs with fp16 operands / fp32 accumulation and fp16 "
                         "activations between the kernels (tiny-cuda-nn's network precision; tables, Adam, warp, FFT and "
                         "losses stay fp32)")
    ap.add_argument("--mlp-fp16", action="store_true", help="same as --precision f16mlp")
    ap.add_argument("--chains", choices=["auto", "fork", "serial"], default="auto",
                    help="image / motion kernel chains as two graph branches (fork), one after the other (serial), or "
                         "decided by the lattice size (auto, the library default)")
    ap.add_argument("--workload", choices=["c2", "c3", "c5"], default="c2",
                    help="c2 (default, the metric's configuration): 320x320, 10 groups; c3: batches of --batch such "
                         "slices on one GPU; c5 (informational): 640x640, 20 groups, implies --table-fp16")
    ap.add_argument("--batch", type=int, default=64, help="slices per batch (c3)")
    ap.add_argument("--lanes", type=int, default=1, help="slices in flight side by side (c3; 1 is fastest)")
    ap.add_argument("--pair", action="store_true", help="c3: two slices per graph, gathers serialised (batch_pair)")
    ap.add_argument("--gather", choices=["all", "rank0"], default="all",
                    help="final images to every rank (all_gather_into_tensor) or to rank 0 only (gather, what the C4 line needs)")
    ap.add_argument("--cpu-iters", type=int, default=30, help="iterations of the cpu_baseline sample (SURVEY 8d: 30)")
    ap.add_argument("--dry-run", action="store_true", help="rank plumbing only, on the CPU with gloo (no GPU, no HIP library)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))      # this process never touches the GPU
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE')} (set by an outer launcher): the two must agree")
    if args.dry_run:
        return dry_run(args)
    args.mlp_fp16 = 2 if args.precision == "bf16x2" else int(bool(args.mlp_fp16 or args.precision == "f16mlp"))
    global H, W, N_MOVEMENTS
    if args.workload == "c5":
        H = W = 640
        N_MOVEMENTS = 20
        args.table_fp16 = True
    B = args.batch if args.workload == "c3" else 1

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world} (set by an outer launcher): the two must agree")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or os.environ.get("BENCH_FORCE_DIST"):
        dist.init_process_group("nccl", device_id=dev)

    import miccai24_immoco_amd as pkg
    from miccai24_immoco_amd import _lib, synth
    from miccai24_immoco_amd.models.immoco import get_solver, lambda_schedule
    from miccai24_immoco_amd.shard import gather_images
    _lib.lib()   # fail loudly without the HIP library

    K, Wm = args.steps, args.warmup
    # ---- synthetic inputs, resident in HBM before the timed region ---------------------------
    slices = []
    n_sl = (Wm + K) * B
    for j in range(n_sl):
        gidx = rank * n_sl + j + 2               # (0 / 1 are the CPU baseline's and the oracle records' slices)
        s = synth.make_slice(H, W, N_MOVEMENTS, gidx, device=dev)     # HIP motion simulator
        masks = pkg.extract_movement_groups(s["lines"], make_list=True)
        slices.append({"kspace": s["kspace"], "masks": masks, "gt": s["gt"].cpu()})
    # the oracle-record slice (data fixture, no oracle code involved) is the first TIMED slice of the metric's workload
    ref_rec, fx = None, os.path.join(ROOT, "tests", "golden")
    if args.workload == "c2" and rank == 0 and args.iters == 3000 and os.path.exists(os.path.join(fx, "c2_slice1_input.npz")):
        import numpy as np
        fin = np.load(os.path.join(fx, "c2_slice1_input.npz"))
        rec = np.load(os.path.join(fx, "c2_oracle_slice1_3000it.npz"))
        kf = torch.from_numpy(fin["kspace"]).to(dev)
        mf = pkg.extract_movement_groups(torch.from_numpy(fin["lines"]).to(dev), make_list=True)
        slices[Wm] = {"kspace": kf, "masks": mf, "gt": synth.phantom(H, W, 1000 + int(fin["slice_idx"]))}
        ref_rec = {"oracle_psnr_db": [round(float(v), 3) for v in rec["oracle_psnr"][:, -1]],
                   "source": "tests/golden/c2_oracle_slice1_3000it.npz (tools/oracle_c2.py, CPU oracle, same input)"}
    nM = int(slices[0]["masks"].shape[0])
    if any(int(sl["masks"].shape[0]) != nM for sl in slices) and args.workload == "c3":
        raise SystemExit("synthetic slices ended up with different group counts")
    def the_solver(lanes=0, grad_parts=None):
        return get_solver(dev, H, W, nM, not args.no_graph, False, args.grad_parts if grad_parts is None else grad_parts,
                          0, args.table_fp16, lanes, mlp_fp16=args.mlp_fp16,
                          serial_chains={"auto": None, "fork": False, "serial": True}[args.chains],
                          batch_pair=bool(args.pair) and B > 1)
    the_solver()      # plans + workspace (one-off, like FFT plan creation)

    def solve(sl):
        return pkg.imcoco_motion_correction(sl["kspace"], sl["masks"], iters=args.iters, learning_rate=1e-2,
                                            lambda_ge=1e-2, use_graph=not args.no_graph, grad_parts=args.grad_parts,
                                            table_fp16=args.table_fp16, mlp_fp16=args.mlp_fp16,
                                            serial_chains={"auto": None, "fork": False, "serial": True}[args.chains])

    def step(j):
        """One step = one pass of the hot path over one batch: a slice (c2, c5) or B slices (c3)."""
        if B == 1:
            return [solve(slices[j])[0]]
        grp = slices[j * B:(j + 1) * B]
        imgs, _ = pkg.imcoco_motion_correction_batch(torch.stack([g["kspace"] for g in grp]), [g["masks"] for g in grp],
                                                     iters=args.iters, lanes=args.lanes, table_fp16=args.table_fp16,
                                                     mlp_fp16=args.mlp_fp16, pair=args.pair, use_graph=not args.no_graph,
                                                     serial_chains={"auto": None, "fork": False, "serial": True}[args.chains])
        return list(imgs)

    def barrier():
        if dist.is_initialized():
            dist.barrier()
    for j in range(Wm):
        step(j)
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    imgs = [im for j in range(K) for im in step(Wm + j)]
    local = torch.stack(imgs)
    allimgs = gather_images(local, K * B * world, dst=0 if args.gather == "rank0" else None)   # the single RCCL collective
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if dist.is_initialized():
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    value = K * B * world / dt
    ms_per_step = dt / K * 1e3
    graph_used = bool(the_solver(args.lanes if B > 1 else 0, 0 if B > 1 else None).graph_active)   # of the timed solves (same solver key)

    out = None
    if rank == 0:
        from miccai24_immoco_amd.utils.evaluate import crop_psnr
        tsl = slices[Wm * B:(Wm + K) * B]
        psnr = [crop_psnr(imgs[j].abs().cpu(), tsl[j]["gt"].abs().cpu()) for j in range(K * B)]
        from miccai24_immoco_amd.utils.data_utils import IFFT
        psnr_in = [crop_psnr(IFFT(tsl[j]["kspace"]).abs().cpu(), tsl[j]["gt"].abs().cpu()) for j in range(K * B)]
        psnr_delta = None
        if ref_rec is not None and K >= 1:
            # PSNR against the reference side (outside the timed region).  The trajectory is chaotic (PSNR oscillates with
            # period 2 by +-1.5 dB, every run goes through a loss blow-up between iterations 1050 and 1460), so only
            # distributions compare: per run the MEDIAN PSNR over a window of iterations, HIP mean minus oracle mean with
            # the standard error of that difference.  Two references, both data fixtures (no oracle code runs here):
            #  * plateau (lambda_GE > 0): iterations 600, 625, ..., 975 against >= 64 draws of the DEVICE oracle (ATen on the
            #    GPU, fp32 atomics; tests/golden/c2_device_oracle_draws.npz, tools/device_oracle_sampler.py): 24 HIP runs;
            #  * end of the solve (2900 ... 2999) against the six full CPU-oracle records: 6 HIP runs.
            from miccai24_immoco_amd.utils.sampling import hip_psnr_samples, summarize, delta_with_se
            from miccai24_immoco_amd.utils.motion_utils import masks_to_col_group as _m2c
            import numpy as np
            rec = np.load(os.path.join(fx, "c2_oracle_slice1_3000it.npz"))
            its = list(rec["oracle_psnr_iters"])
            end_it, pl_it = [2900, 2925, 2950, 2975, 2999], list(range(600, 1000, 25))
            o_end = np.median(rec["oracle_psnr"][:, [its.index(t) for t in end_it]], axis=1)
            o_last = rec["oracle_psnr"][:, -1]
            ksl = tsl[0]["kspace"]
            kin1, cg1, gt1 = ksl / ksl.abs().max() * 16000, _m2c(tsl[0]["masks"]), tsl[0]["gt"].abs().cpu()
            h_end, h_last, h_pl = [], [psnr[0]], []
            for _ in range(6):
                ps_, _l = hip_psnr_samples(the_solver(), kin1, cg1, gt1, 3000, pl_it + end_it)
                h_pl.append(float(np.median([ps_[t] for t in pl_it])))
                h_end.append(float(np.median([ps_[t] for t in end_it])))
                h_last.append(ps_[2999])
            plateau = None
            dpath = os.path.join(fx, "c2_device_oracle_draws.npz")
            if os.path.exists(dpath):
                dd = np.load(dpath)
                if "s1_plateau_psnr" in dd:
                    pits = list(dd["psnr_plateau_iters"])
                    o_pl = np.median(dd["s1_plateau_psnr"][:, [pits.index(t) for t in pl_it]].astype(np.float64), axis=1)
                    for _ in range(18):
                        ps_, _l = hip_psnr_samples(the_solver(), kin1, cg1, gt1, 3000, pl_it)
                        h_pl.append(float(np.median([ps_[t] for t in pl_it])))
                    d_pl = delta_with_se(h_pl, o_pl)
                    plateau = {"delta_db": round(d_pl[0], 3), "se_db": round(d_pl[1], 3), "variance_ratio": round(d_pl[2], 2),
                               "n_hip_runs": len(h_pl), "n_oracle_draws": int(len(o_pl)),
                               "hip_mean_db": round(float(np.mean(h_pl)), 3), "oracle_mean_db": round(float(o_pl.mean()), 3),
                               "low_runs_below_38db": {"hip": int(sum(v < 38 for v in h_pl)), "oracle": int((o_pl < 38).sum())},
                               "what": "per run: median PSNR over iterations 600, 625, ..., 975 (lambda_GE = 1e-2); oracle = "
                                       "device-oracle draws (fp32, ATen atomics) of tests/golden/c2_device_oracle_draws.npz"}
            d_end, d_last = delta_with_se(h_end, o_end), delta_with_se(h_last, o_last)
            psnr_delta = {"psnr_delta_db": plateau["delta_db"] if plateau else round(d_end[0], 3),
                          "psnr_delta_se_db": plateau["se_db"] if plateau else round(d_end[1], 3),
                          "psnr_delta_is": "plateau_600_975 (device-oracle draws)" if plateau else "end_window_median (CPU records)",
                          "slice": "config C2, slice 1", "plateau_600_975": plateau,
                          "end_window_median": {"delta_db": round(d_end[0], 3), "se_db": round(d_end[1], 3),
                                                "variance_ratio": round(d_end[2], 2), "n_hip_runs": len(h_end),
                                                "n_oracle_records": int(len(o_end)),
                                                "what": "per run: median PSNR over 2900, 2925, ..., 2999 (lambda_GE = 0) against the six "
                                                        "full CPU-oracle records (tests/golden/c2_oracle_slice1_3000it.npz)"},
                          "final_forward": {"delta_db": round(d_last[0], 3), "se_db": round(d_last[1], 3), "n_hip_runs": len(h_last),
                                            "hip_timed_run_psnr_db": round(psnr[0], 3)},
                          "reference_script_iters_200": "profiles/r04_cells_vs_device_oracle.txt: slices 1 / 4 / 9, 64 HIP runs against "
                                                        "64 device-oracle draws each (the suite's test_cells_vs_device_oracle_draws re-measures them)",
                          "note": "HIP mean minus oracle mean with the standard error of that difference (both samples)"}
        # ---- roofline: per-kernel device time with HIP events on the solver's stream ----------
        solver = the_solver()
        sl = slices[0]
        k = sl["kspace"]
        kin = k / k.abs().max() * 16000
        from miccai24_immoco_amd.utils.motion_utils import masks_to_col_group
        cg = masks_to_col_group(sl["masks"])
        pi, pm = solver.init_params()
        ai = torch.zeros(2 * pi.numel(), device=dev)
        am = torch.zeros(2 * pm.numel(), device=dev)
        # live duration of the dominant kernel under the run's own concurrency: a short EAGER solve on the
        # same two streams (HIP reports no elapsed time for events recorded as graph nodes)
        solver.set_graph(False)
        dom = []
        for _ in range(5):
            solver.solve(kin, cg, pi, pm, ai, am, 10, 1e-2, lambda_schedule(3000, 1e-2)[:10])
            dom.append(solver.dominant_kernel_ms)
        solver.set_graph(not args.no_graph)
        dom_ms = sorted(dom)[len(dom) // 2]
        solver.profile(kin, cg, pi, pm, ai, am, reps=3)            # warm
        phases = solver.profile(kin, cg, pi, pm, ai, am, reps=10)
        ab = algorithmic_bytes(solver, nM)
        t_iter_ms = sum(ms for _, ms in phases)
        # Dominant kernel: the motion grid's encode backward, timed with HIP events on the solver's stream.
        # `kernel_ms_isolated` is its average over a serial eager pass of the iteration (immoco_solver_profile);
        # `kernel_ms` = `kernel_ms_concurrent` is the same kernel between event markers in an EAGER pass with the image-INR
        # chain beside it on the second stream - the conditions of the replayed graph, and the duration rocprofv3's kernel
        # stats of this command report.
        name = "motion_encode_bwd"
        ms = dict(phases)[name]
        ms_conc = dom_ms if dom_ms > 0 else ms
        # `achieved` is priced at the duration the kernel has IN THE RUN (beside the image chain on the second stream): that is
        # what rocprofv3's kernel stats of this command average (profiles/r04_kernel_stats_default_bench.csv: 0.534 ms; here
        # 0.53) - in round 4 the whole image chain shares the chip with it, so it is well above the isolated 0.43 ms.  The
        # isolated figure is reported beside it (`frac_isolated`).
        achieved = ab[name] / (ms_conc * 1e-3) / 1e9
        achieved_iso = ab[name] / (ms * 1e-3) / 1e9
        # 28 B per parameter (30 with the fp16 shadow write of config 5) + 8 B per pixel
        b_iter = (30 if args.table_fp16 else 28) * (solver.n_params_image + solver.n_params_motion) + 8 * H * W
        iter_ms_graph = ms_per_step / args.iters / B
        traffic, traffic_src = None, None
        tsuffix = ("_c5" if args.workload == "c5" else "") + ({0: "", 1: "_f16mlp", 2: "_bf16x2"}[int(args.mlp_fp16)]) + ".json"
        tpath = next((q for q in (os.path.join(ROOT, "profiles", r + "_traffic" + tsuffix) for r in ("r04", "r03"))
                      if os.path.exists(q)), None)
        if tpath is None:
            traffic_src = "no stored --pmc traffic file for this mode (profiles/r0N_traffic" + tsuffix + ")"
        if tpath is not None:
            try:
                tj = json.load(open(tpath))
                traffic = tj["kernels"][name]["hbm_bytes_corrected"]
                traffic_src = os.path.relpath(tpath, ROOT) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command)"
            except Exception:
                traffic = None
        # The same request shape measured bare on this GPU (csrc/probe.hip): the kernel issues one divergent
        # 8-byte dL/denc gather per (point, corner, level) entry, inside a 2 MB footprint per XCD (one part of
        # a level slice); the probe issues random aligned 8-byte loads inside 2 MB and nothing else.
        import ctypes as C
        from miccai24_immoco_amd import _lib as L
        probe_ms = C.c_float()
        L.check(L.lib().immoco_probe_gather(2 << 20, 8, 1024000, 64, 5,
                                            C.c_void_p(torch.cuda.current_stream().cuda_stream),
                                            C.byref(probe_ms)), "probe_gather")
        ceiling = 1024000 * 64 / (probe_ms.value * 1e-3) / 1e9
        n_alg = nM * H * W * 16 * 8                    # one contribution per (point, corner, level)
        n_req = int(L.lib().immoco_solver_plan_entries(solver.handle, 1)) or n_alg   # gathers issued (twins once)
        gather = {"contributions_per_launch": n_alg, "requests_per_launch": n_req,
                  "achieved_Greq_s": round(n_req / (ms * 1e-3) / 1e9, 1),
                  "achieved_concurrent_Greq_s": round(n_req / (ms_conc * 1e-3) / 1e9, 1),
                  "ceiling_Greq_s": round(ceiling, 1), "frac": round(n_req / (ms * 1e-3) / 1e9 / ceiling, 4),
                  "frac_concurrent": round(n_req / (ms_conc * 1e-3) / 1e9 / ceiling, 4),
                  "ceiling_source": "immoco_probe_gather, measured in this run: random aligned 8-byte loads, "
                                    "2 MB footprint, 1 024 000 lanes x 64 loads"}
        roofline = {
            "bound": "hbm", "kernel": name, "achieved": round(achieved, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": round(achieved / PEAK_HBM_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
            "traffic_measured_in_this_run": False,
            "kernel_ms": round(ms_conc, 4), "kernel_ms_isolated": round(ms, 4), "kernel_ms_concurrent": round(ms_conc, 4),
            "achieved_isolated": round(achieved_iso, 2), "frac_isolated": round(achieved_iso / PEAK_HBM_GBS, 5),
            "kernel_algorithmic_bytes": ab[name],
            "note": "gather kernels are bound by the rate of divergent cache-line requests (rocprof: TA busy 94 %), "
                    "not by HBM bytes: see `gather` for the measured ceiling of that request shape",
            "gather": gather,
            "iteration": {"algorithmic_bytes": b_iter, "ms_graph": round(iter_ms_graph, 4),
                          "ms_sum_of_kernels_eager": round(t_iter_ms, 4),
                          "achieved_GBs": round(b_iter / (iter_ms_graph * 1e-3) / 1e9, 2),
                          "frac": round(b_iter / (iter_ms_graph * 1e-3) / 1e9 / PEAK_HBM_GBS, 5)},
            "kernels_ms_isolated": {n: round(m, 4) for n, m in phases},
        }
        out = {
            "metric": "slices/sec at 320x320, 10 motion groups, 3000 iters; PSNR delta vs ref" if args.workload != "c5"
            else "slices/sec at 640x640, 20 motion groups (BASELINE config 5; informational)",
            "value": round(value, 5), "unit": "slices/s", "n_gpus": world,
            "n_ranks_seen": dist.get_world_size() if dist.is_initialized() else 1, "steps": K, "warmup": Wm,
            "ms_per_step": round(ms_per_step, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": {(False, 0): "f32", (True, 0): "f32+f16tab", (False, 1): "f16mlp/f32acc", (True, 1): "f16mlp+f16tab/f32acc",
                      (False, 2): "f32 (MLP products: bf16x2 split, f32 accumulate)",
                      (True, 2): "f32+f16tab (MLP products: bf16x2 split)"}[(bool(args.table_fp16), int(args.mlp_fp16))],
            "data": "synthetic",
            "config": {"workload": {"c2": "C2: single 320x320 slice, 10 motion groups, 3000 Adam iters, hash-grid INRs",
                                    "c3": f"C3: batch of {B} independent 320x320 slices on one GPU, {args.lanes} in flight",
                                    "c5": "C5: single 640x640 slice, 20 motion groups, fp16 hash-grid features + fp32 Adam"
                                    }[args.workload],
                       "H": H, "W": W, "motion_groups": nM, "iters": args.iters, "slices_per_gpu": K * B,
                       "graph": graph_used, "parallelism": f"slices sharded over {world} GPU(s)"},
            "psnr_db": {"solved": [round(p, 3) for p in psnr], "corrupted_input": [round(p, 3) for p in psnr_in]},
            "psnr_delta_vs_ref": psnr_delta,
            "roofline": roofline,
        }
        if world == 1 and args.workload == "c2" and not args.no_alt_precision:
            # the other MLP arithmetics on the same slices, outside the timed region (2 slices each): informational lines
            labels = {0: "f32", 1: "f16mlp/f32acc", 2: "f32 (MLP products: bf16x2 split, f32 accumulate)"}
            notes = {0: "exact fp32 everywhere",
                     1: "both MLPs with fp16 operands, fp32 accumulation, fp16 activations between the kernels: tiny-cuda-nn's "
                        "network precision (/root/reference/src/models/immoco.py:11-25,60-65); PSNR statistics within +-0.5 dB of fp32 "
                        "over eight initialisations, up to -1.4 dB at 200 iterations from the reference's one (DESIGN.md 2.2, 2.3)",
                     2: "every MLP matrix operand split into two bf16 terms (product error <= 2^-16.5), everything else fp32 "
                        "(DESIGN.md 4.3)"}
            out["other_precision"] = []
            n_alt = min(2, K)
            for mode in (0, 1, 2):
                if mode == int(args.mlp_fp16):
                    continue
                get_solver(dev, H, W, nM, not args.no_graph, False, args.grad_parts, 0, args.table_fp16, 0, mlp_fp16=mode)
                torch.cuda.synchronize()
                ta = time.perf_counter()
                alt_imgs = [pkg.imcoco_motion_correction(tsl[j]["kspace"], tsl[j]["masks"], iters=args.iters, learning_rate=1e-2,
                                                         lambda_ge=1e-2, use_graph=not args.no_graph, grad_parts=args.grad_parts,
                                                         table_fp16=args.table_fp16, mlp_fp16=mode)[0] for j in range(n_alt)]
                torch.cuda.synchronize()
                dta = time.perf_counter() - ta
                out["other_precision"].append({
                    "dtype": labels[mode], "slices": n_alt, "value": round(n_alt / dta, 5), "unit": "slices/s",
                    "ms_per_iteration": round(dta / n_alt / args.iters * 1e3, 4),
                    "psnr_db": [round(crop_psnr(alt_imgs[j].abs().cpu(), tsl[j]["gt"].abs().cpu()), 3) for j in range(n_alt)],
                    "note": notes[mode] + "; same slices as the timed run, outside the timed region"})
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_iters)
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
