#!/usr/bin/env python
"""The narrow MLP backward taken apart (csrc/mlp_mfma.hip: mlp_bwd64_pipe_kernel and its ablations, IMMOCO_MLP_RAWLOAD;
diagnostics library only; results of round 4 in profiles/r04_narrow_mlp_bwd_ablations.txt).

    make -C miccai24_immoco_amd/csrc diag
    IMMOCO_LIB_PATH=$PWD/miccai24_immoco_amd/csrc/libimmoco_hip_diag.so [IMMOCO_MLP_BWD64=pipe|pipe_partials|pipe_nomfma|...] \
        [IMMOCO_MLP_RAWLOAD=1] [CHECK_ACT=ReLU] [CHECK_LAYOUT=point] python tools/check_pipe_bwd.py --quick
            one process: the selected kernel against a float64 torch evaluation at n = 20 000 and 1 024 000, then timed
    python tools/check_pipe_bwd.py
            two processes (the switch is read once per process): shipped kernel against IMMOCO_MLP_BWD64=pipe - d enc must be
            bit-identical, dW1 / dW2 equal to the atomics' rounding (not run in round 4)

Child mode (internal): python tools/check_pipe_bwd.py --child out.npz"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIAG = os.path.join(ROOT, "miccai24_immoco_amd", "csrc", "libimmoco_hip_diag.so")


def child(out):
    sys.path.insert(0, ROOT)
    import ctypes as C
    import numpy as np
    import torch
    from miccai24_immoco_amd import _lib as L
    dev = torch.device("cuda", 0)
    res = {}
    for n in (64, 1000, 20000, 1024000):
        g = torch.Generator().manual_seed(n)
        cfg = L.mlp_cfg(32, 2, {"otype": "FullyFusedMLP", "activation": "Tanh", "output_activation": "None",
                                   "n_neurons": 64, "n_hidden_layers": 1})   # the motion net (immoco.py:36-42)
        hid, pad = cfg.n_hidden, cfg.n_out_padded
        x = (torch.randn(16, n, 2, generator=g) * 0.5).to(dev)           # level-major encoding, as in the solver
        w1 = (torch.randn(hid, 32, generator=g) * 0.3).to(dev)
        w2 = (torch.randn(pad, hid, generator=g) * 0.3).to(dev)
        dd = torch.randn(n, 2, generator=g).to(dev)
        dw1, dw2 = torch.zeros(hid * 32, device=dev), torch.zeros(pad * hid, device=dev)
        xl = x.clone()
        st = L.stream_ptr()
        L.check(L.lib().immoco_mlp_bwd(C.byref(cfg), L.ptr(xl), 2, 2 * n, n, L.ptr(w1), L.ptr(w2), L.ptr(dd), L.ptr(xl),
                                       L.ptr(dw1), L.ptr(dw2), st), "mlp_bwd")
        torch.cuda.synchronize()
        res[f"denc_{n}"], res[f"dw1_{n}"], res[f"dw2_{n}"] = xl.cpu().numpy(), dw1.cpu().numpy(), dw2.cpu().numpy()
        if n == 1024000:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                xl.copy_(x)
                L.check(L.lib().immoco_mlp_bwd(C.byref(cfg), L.ptr(xl), 2, 2 * n, n, L.ptr(w1), L.ptr(w2), L.ptr(dd), L.ptr(xl),
                                               L.ptr(dw1), L.ptr(dw2), st), "mlp_bwd")
            e1.record()
            torch.cuda.synchronize()
            t_all = e0.elapsed_time(e1) / 20
            e0.record()
            for _ in range(20):
                xl.copy_(x)
            e1.record()
            torch.cuda.synchronize()
            res["ms"] = np.float64(t_all - e0.elapsed_time(e1) / 20)
    np.savez(out, **res)


def quick():
    """One process, pipelined kernel only (IMMOCO_MLP_BWD64=pipe must be set, diagnostics library): against a float64 torch
    evaluation at n = 20 000 and timed at n = 1 024 000."""
    sys.path.insert(0, ROOT)
    import ctypes as C
    import torch
    from miccai24_immoco_amd import _lib as L
    assert os.environ.get("IMMOCO_LIB_PATH", "").endswith("_diag.so")
    act = os.environ.get("CHECK_ACT", "Tanh")
    print("kernel:", os.environ.get("IMMOCO_MLP_BWD64", "shipped"), "activation:", act, flush=True)
    dev = torch.device("cuda", 0)
    cfg = L.mlp_cfg(32, 2, {"otype": "FullyFusedMLP", "activation": act, "output_activation": "None",
                            "n_neurons": 64, "n_hidden_layers": 1})
    hid, pad = cfg.n_hidden, cfg.n_out_padded
    for n in (20000, 1024000):
        g = torch.Generator().manual_seed(n)
        x = (torch.randn(16, n, 2, generator=g) * 0.5).to(dev)
        w1 = (torch.randn(hid, 32, generator=g) * 0.3).to(dev)
        w2 = (torch.randn(pad, hid, generator=g) * 0.3).to(dev)
        dd = torch.randn(n, 2, generator=g).to(dev)
        dw1, dw2 = torch.zeros(16 * hid * 32, device=dev), torch.zeros(16 * pad * hid, device=dev)   # 16 copies: the spread ablations
        xl = x.clone()
        st = L.stream_ptr()
        L.check(L.lib().immoco_mlp_bwd(C.byref(cfg), L.ptr(xl), 2, 2 * n, n, L.ptr(w1), L.ptr(w2), L.ptr(dd), L.ptr(xl),
                                       L.ptr(dw1), L.ptr(dw2), st), "mlp_bwd")
        torch.cuda.synchronize()
        dw1s, dw2s = dw1.view(16, hid * 32).sum(0), dw2.view(16, pad * hid).sum(0)
        enc = x.permute(1, 0, 2).reshape(n, 32).double()
        hh = torch.tanh(enc @ w1.double().t()) if act == "Tanh" else torch.relu(enc @ w1.double().t())
        dpre = (dd.double() @ w2[:2].double()) * ((1.0 - hh * hh) if act == "Tanh" else (hh > 0).double())
        denc = (dpre @ w1.double()).reshape(n, 16, 2).permute(1, 0, 2)
        rd = float((xl.double() - denc).abs().max() / denc.abs().max())
        r1 = float((dw1s.double().view(hid, 32) - dpre.t() @ enc).abs().max() / (dpre.t() @ enc).abs().max())
        r2 = float((dw2s.double().view(pad, hid)[:2] - dd.double().t() @ hh).abs().max() / (dd.double().t() @ hh).abs().max())
        print(f"n {n}: d enc max err / max {rd:.2e}, dW1 {r1:.2e}, dW2 {r2:.2e}", flush=True)
        if n == 1024000:
            ps_, ls_ = (32, 2) if os.environ.get("CHECK_LAYOUT") == "point" else (2, 2 * n)   # timing only: same bytes, point-major
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            ev[0].record()
            for _ in range(20):
                xl.copy_(x)
                L.check(L.lib().immoco_mlp_bwd(C.byref(cfg), L.ptr(xl), ps_, ls_, n, L.ptr(w1), L.ptr(w2), L.ptr(dd), L.ptr(xl),
                                               L.ptr(dw1), L.ptr(dw2), st), "mlp_bwd")
            ev[1].record()
            for _ in range(20):
                xl.copy_(x)
            ev[2].record()
            torch.cuda.synchronize()
            print(("point-major " if ps_ == 32 else "level-major ") + "kernel %.4f ms (shipped tanh kernel in bench.py kernels_ms_isolated: 0.176); the 131 MB + 131 MB copy beside it: %.4f ms" %
                  ((ev[0].elapsed_time(ev[1]) - ev[1].elapsed_time(ev[2])) / 20, ev[1].elapsed_time(ev[2]) / 20), flush=True)


def main():
    import numpy as np
    assert os.path.exists(DIAG), "build the diagnostics library first: make -C miccai24_immoco_amd/csrc diag"
    outs = {}
    for name, extra in (("shipped", {}), ("pipe", {"IMMOCO_MLP_BWD64": "pipe"})):
        out = f"/tmp/check_pipe_{name}.npz"
        env = dict(os.environ, IMMOCO_LIB_PATH=DIAG, **extra)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", out], check=True, env=env, timeout=600)
        outs[name] = np.load(out)
    a, b = outs["shipped"], outs["pipe"]
    ok = True
    for k in sorted(a.files):
        if k == "ms":
            continue
        if k.startswith("denc"):
            same = bool(np.array_equal(a[k], b[k]))
            print(k, "bit-identical" if same else f"DIFFERS: max abs {np.abs(a[k] - b[k]).max():.3e}")
            ok &= same
        else:
            rel = float(np.abs(a[k] - b[k]).max() / max(np.abs(a[k]).max(), 1e-30))
            print(k, f"max abs difference / max abs {rel:.2e}")
            ok &= rel <= 1e-5
    print("kernel time at 1 024 000 points: shipped %.4f ms, pipelined %.4f ms" % (float(a["ms"]), float(b["ms"])))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--child":
        child(sys.argv[2])
    elif len(sys.argv) == 2 and sys.argv[1] == "--quick":
        quick()
    else:
        main()
