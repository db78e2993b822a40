"""Teacher-forced ONE-step comparison on any C2 slice / schedule / iteration (GPU box): the device oracle runs K iterations,
its state goes to the CPU oracle, the device oracle and HIP for iteration K; loss and gradient differences are printed.
    python tools/diag_tf_slice.py <slice> <sched_iters> <K> [K ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
torch.set_num_threads(16)
import miccai24_immoco_amd as pkg
from miccai24_immoco_amd import _lib as L
from oracle import immoco_oracle as orc, synth_cpu
import test_gpu_ops as T
sl, sched = int(sys.argv[1]), int(sys.argv[2])
s = synth_cpu.make_slice(320, 320, 10, sl)
masks = orc.extract_movement_groups(s["lines"], make_list=True)
print("slice", sl, "groups", masks.shape[0], "lines", int(s["lines"].sum()))
for K in [int(x) for x in sys.argv[3:]]:
    st = T._device_oracle_state(orc, s["kspace"], masks, sched, K)
    rec = T._oracle_step_from_state(orc, st, masks, K)
    try:
        rep = T._teacher_forced_step(pkg, L, orc, s["kspace"], masks, sched, K, record=rec, loss_rtol=1.0, image_tol=1.0, still_tol=1.0)
        print("K", K, {k: (v if not isinstance(v, dict) else {a: v[a] for a in ("grad_rel_l2", "grad_max_abs_over_max", "upd_rel_l2", "upd_max")}) for k, v in rep.items() if k != "_oracle_record"})
    except AssertionError as e:
        print("K", K, "assert", str(e)[:500])
