#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE's own Python
(/root/reference/src) in the build container.  Outputs: tests/golden/*.npz.

Only data (inputs / expected outputs) is written; no reference source or
bytecode is copied.  The reference hard-requires modules that do not exist
here, so this generator registers inert stand-ins before importing it:

* ``h5py``, ``seaborn``, ``piq`` — imported at module top but unused on the
  paths exercised (data_utils.py:1, evaluate.py:13-15);
* ``IPython.display`` — only used when debug=True (immoco.py:5);
* ``tinycudann`` — CUDA-only third-party INR (immoco.py:1).  The stand-in is
  the oracle's fp32 hash-grid INR (oracle/immoco_oracle.py:OracleINR) so that
  the *reference's* IMMoCo.forward and imcoco_motion_correction loop drive it;
  the INR arithmetic itself is therefore NOT pinned by these vectors
  (parity unpinned against tiny-cuda-nn, see oracle header);
* ``torch.Tensor.cuda`` is made the identity (immoco.py:141 hard-calls it).

Run:  python tools/gen_golden.py   (needs /root/reference; CPU only)
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("IMMOCO_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")


def _install_stubs():
    from oracle import immoco_oracle as orc

    for name in ("h5py", "seaborn"):
        sys.modules.setdefault(name, types.ModuleType(name))
    piq = types.ModuleType("piq")
    piq.ssim = piq.haarpsi = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("piq absent"))
    sys.modules.setdefault("piq", piq)
    ip = types.ModuleType("IPython")
    ipd = types.ModuleType("IPython.display")
    ipd.clear_output = ipd.display = lambda *a, **k: None
    ip.display = ipd
    sys.modules.setdefault("IPython", ip)
    sys.modules.setdefault("IPython.display", ipd)
    tcnn = types.ModuleType("tinycudann")
    tcnn.NetworkWithInputEncoding = orc.OracleINR
    sys.modules["tinycudann"] = tcnn
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.empty_cache = lambda: None
    sys.path.insert(0, os.path.join(REF, "src"))


def phantom(H, W, seed):
    """Small analytic complex phantom (ellipses + smooth phase)."""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
    img = torch.zeros(H, W)
    for _ in range(6):
        cx, cy = (torch.rand(2, generator=g) - 0.5).tolist()
        a, b = (0.15 + 0.5 * torch.rand(2, generator=g)).tolist()
        v = 0.1 + 0.9 * torch.rand(1, generator=g).item()
        img = img + v * (((xx - cx) / a) ** 2 + ((yy - cy) / b) ** 2 <= 1).float()
    phase = 0.6 * (xx * xx - 0.5 * yy * xx)
    return (img * torch.exp(1j * phase)).to(torch.complex64)


def main():
    _install_stubs()
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)

    from models import immoco as ref_immoco          # reference
    from utils import data_utils as ref_du           # reference
    from utils import evaluate as ref_ev             # reference
    from utils import losses as ref_losses           # reference
    from utils import motion_utils as ref_mu         # reference

    # ---------------- A. operators ------------------------------------------------
    ops = {}
    g = torch.Generator().manual_seed(7)
    for tag, shape in (("even", (3, 16, 20)), ("mod2", (2, 10, 14)), ("odd", (2, 5, 7))):
        x = torch.complex(torch.randn(shape, generator=g), torch.randn(shape, generator=g))
        ops[f"fft_{tag}_in"] = x.numpy()
        ops[f"fft_{tag}_out"] = ref_du.FFT(x).numpy()
        ops[f"ifft_{tag}_out"] = ref_du.IFFT(x).numpy()
    x = torch.complex(torch.randn(16, 20, generator=g), torch.randn(16, 20, generator=g)).requires_grad_(True)
    x.data[3, 4] = x.data[3, 5]          # exercise the g=0 corner of the entropy
    x.data[4, 4] = x.data[3, 4]
    loss = ref_losses.GradientEntropyLoss()(x)
    loss.backward()
    ops["ge_in"] = x.detach().numpy()
    ops["ge_loss"] = np.float32(loss.item())
    ops["ge_grad"] = x.grad.numpy()
    ops["make_grids_2_3_4"] = ref_immoco.make_grids((2, 3, 4)).numpy()
    ops["make_grids_1_3_5"] = ref_immoco.make_grids((1, 3, 5)).numpy()
    # evaluate.py metrics
    a = torch.rand(1, 1, 12, 12, generator=g) * 3 + 1
    b = torch.rand(1, 1, 12, 12, generator=g)
    ops["metric_a"], ops["metric_b"] = a.numpy(), b.numpy()
    ops["normalize_a"] = ref_ev.normalize(a).numpy()
    ops["psnr_ab"] = np.float32(ref_ev.my_psnr(ref_ev.normalize(a), ref_ev.normalize(b), data_range=1.0))
    ops["rmse_ab"] = np.float32(ref_ev.rmse(ref_ev.normalize(a), ref_ev.normalize(b)))
    # gradients THROUGH the reference's FFT / IFFT (torch autograd) for odd and even sizes: pins the adjoints
    # of the package's centred transforms (odd sizes: fftshift and ifftshift are different rolls)
    g2 = torch.Generator().manual_seed(23)
    for tag, shape in (("odd", (2, 5, 7)), ("even", (2, 6, 8))):
        x = torch.complex(torch.randn(shape, generator=g2), torch.randn(shape, generator=g2)).requires_grad_(True)
        y = torch.complex(torch.randn(shape, generator=g2), torch.randn(shape, generator=g2))
        ops[f"adj_{tag}_x"], ops[f"adj_{tag}_y"] = x.detach().numpy(), y.numpy()
        for name, fn in (("fft", ref_du.FFT), ("ifft", ref_du.IFFT)):
            x.grad = None
            (torch.view_as_real(fn(x)) * torch.view_as_real(y)).sum().backward()
            ops[f"adj_{tag}_{name}_grad"] = x.grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "ops.npz"), **ops)

    # ---------------- B. line-select masks (bit-exact integer path) ----------------
    mk = {}
    vecs = {
        "typical": [0, 0, 1, 1, 1, 0, 0, 1, 0, 0, 0, 1, 1, 0, 0, 0],
        "last_true": [0, 1, 0, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 1, 1],
        "first_true": [1, 1, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0],
        "single": [0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0],
        "all_true": [1] * 12,
        "alternating": [1, 0] * 8,
    }
    gg = torch.Generator().manual_seed(11)
    vecs["random320"] = (torch.rand(320, generator=gg) < 0.15).int().tolist()
    for k, v in vecs.items():
        t = torch.tensor(v).bool()
        mk[f"{k}_vec"] = np.array(v, dtype=np.uint8)
        mk[f"{k}_groups"] = ref_mu.extract_movement_groups(t, make_list=False).numpy()[0].astype(np.int32)
        ml = ref_mu.extract_movement_groups(t, make_list=True)
        # masks are constant along rows: store row 0 only + the shape
        assert bool((ml == ml[:, :1, :]).all())
        mk[f"{k}_list_row0"] = ml[:, 0, :].numpy().astype(np.uint8)
        mk[f"{k}_list_shape"] = np.array(ml.shape, dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "masks.npz"), **mk)

    # ---------------- C. motion simulator ------------------------------------------
    sim = {}
    for tag, (H, nm, seed) in {"s32": (32, 3, 5), "s64": (64, 5, 9)}.items():
        img = phantom(H, H, seed)
        torch.manual_seed(seed)
        ksp, mask, rot, tr = ref_mu.motion_simulation2D(img.clone(), n_movements=nm)
        sim[f"{tag}_img"] = img.numpy()
        sim[f"{tag}_seed"], sim[f"{tag}_nm"] = np.int64(seed), np.int64(nm)
        sim[f"{tag}_ksp"] = ksp.numpy()
        sim[f"{tag}_mask_row0"] = mask[0].numpy().astype(np.uint8)
        sim[f"{tag}_rot"], sim[f"{tag}_tr"] = rot.numpy(), tr.numpy()
    np.savez_compressed(os.path.join(OUT, "motion_sim.npz"), **sim)

    # ---------------- D. forward operator + solver loop ----------------------------
    # The reference's IMMoCo / imcoco_motion_correction with the oracle INR plugged in
    # as `tinycudann` (seed 1337 for both INRs, the upstream default).
    sol = {}
    for tag, (H, nm, iters, seed) in {"c32": (32, 2, 20, 3), "c48": (48, 3, 30, 4)}.items():
        img = phantom(H, H, seed)
        torch.manual_seed(seed)
        ksp, mask, _, _ = ref_mu.motion_simulation2D(img.clone(), n_movements=nm)
        masks = ref_mu.extract_movement_groups(mask.sum(0).div(H) > 0.2, make_list=True)
        model = ref_immoco.IMMoCo(masks)
        with torch.no_grad():
            k0, im0 = model()
        sol[f"{tag}_gt"] = img.numpy()
        sol[f"{tag}_ksp"] = ksp.numpy()
        sol[f"{tag}_masks_row0"] = masks[:, 0, :].numpy().astype(np.uint8)
        sol[f"{tag}_iters"] = np.int64(iters)
        sol[f"{tag}_fwd0_kspace"] = k0.numpy()
        sol[f"{tag}_fwd0_image"] = im0.numpy()
        sol[f"{tag}_identy_grid"] = model.identy_grid.numpy()
        sol[f"{tag}_input_grid"] = model.input_grid.numpy()
        image_prior, kfm = ref_immoco.imcoco_motion_correction(
            ksp, masks, iters=iters, learning_rate=1e-2, lambda_ge=1e-2, debug=False)
        sol[f"{tag}_image_prior"] = image_prior.detach().numpy()
        sol[f"{tag}_kfm"] = kfm.detach().numpy()
        print(tag, "done; |image| max", float(image_prior.abs().max()))
    np.savez_compressed(os.path.join(OUT, "solver.npz"), **sol)
    # ---------------- E. Autofocusing baseline (src/models/autofocusing.py, loop of
    # src/test/test_autofocusing.py:66-74) - pure torch in the reference, fully pinned ----------
    from models import autofocusing as ref_af      # reference
    af = {}
    for tag, (H, nm, seed) in {"a32": (32, 2, 13), "a48": (48, 3, 17)}.items():
        img = phantom(H, H, seed)
        torch.manual_seed(seed)
        ksp, mask, _, _ = ref_mu.motion_simulation2D(img.clone(), n_movements=nm)
        masks = ref_mu.extract_movement_groups(mask.sum(0).div(H) > 0.2, make_list=True)
        scale = ref_du.IFFT(ksp).abs().max()
        ksp = ksp / scale
        model = ref_af.Autofocusing(masks)
        nM = masks.shape[0]
        with torch.no_grad():
            model.motion_parameters["rot_vector"].copy_(torch.linspace(-3.0, 4.0, nM))
            model.motion_parameters["x_shifts"].copy_(torch.linspace(2.0, -5.0, nM))
            model.motion_parameters["y_shifts"].copy_(torch.linspace(-1.5, 3.5, nM))
        kout = model(ksp)
        loss = ref_losses.GradientEntropyLoss()(ref_du.IFFT(kout)) * 1e-4
        loss.backward()
        af[f"{tag}_ksp"] = ksp.numpy()
        af[f"{tag}_masks_row0"] = masks[:, 0, :].numpy().astype(np.uint8)
        af[f"{tag}_kout"] = kout.detach().numpy()
        af[f"{tag}_loss"] = np.float32(loss.item())
        for nme in ("rot_vector", "x_shifts", "y_shifts"):
            af[f"{tag}_p_{nme}"] = model.motion_parameters[nme].detach().numpy().copy()
            af[f"{tag}_g_{nme}"] = model.motion_parameters[nme].grad.numpy().copy()
        # the optimisation loop of test_autofocusing.py:66-74 from zero parameters
        model = ref_af.Autofocusing(masks)
        opt = torch.optim.Adam(model.parameters(), lr=1.0)
        hist = []
        for i in range(12):
            opt.zero_grad()
            kr = model(ksp)
            l = ref_losses.GradientEntropyLoss()(ref_du.IFFT(kr)) * 1e-4
            l.backward()
            opt.step()
            hist.append(l.item())
        af[f"{tag}_loop_loss"] = np.array(hist, dtype=np.float32)
        af[f"{tag}_loop_kout"] = kr.detach().numpy()
        for nme in ("rot_vector", "x_shifts", "y_shifts"):
            af[f"{tag}_loop_{nme}"] = model.motion_parameters[nme].detach().numpy().copy()
    np.savez_compressed(os.path.join(OUT, "autofocus.npz"), **af)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)), "bytes")


if __name__ == "__main__":
    main()
