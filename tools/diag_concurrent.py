"""Diagnostic: throughput with 1, 2, 3 slices in flight on one GPU (independent solver handles)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import miccai24_immoco_amd as pkg
from miccai24_immoco_amd import synth
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 600
sl = []
for i in range(6):
    s = synth.make_slice(320, 320, 10, i, device="cuda")
    sl.append((s["kspace"], pkg.extract_movement_groups(s["lines"], make_list=True)))
streams = [torch.cuda.Stream() for _ in range(3)]
for conc in (1, 2, 3):
    for w in range(conc):
        with torch.cuda.stream(streams[w]):
            pkg.imcoco_motion_correction(sl[w][0], sl[w][1], iters=20, instance=w)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = []
    for j in range(6):
        with torch.cuda.stream(streams[j % conc]):      # one caller stream per handle: solves overlap
            outs.append(pkg.imcoco_motion_correction(sl[j][0], sl[j][1], iters=iters, instance=j % conc)[0])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"in flight {conc}: {6 / dt:.3f} slices/s at {iters} it  ({dt / 6 * 1e3 / iters:.3f} ms per slice-iteration)", flush=True)
