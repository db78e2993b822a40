"""Known-answer tests pinning the oracle's restatement of the tiny-cuda-nn hash
grid (parity vs tiny-cuda-nn itself is unpinned)."""
import numpy as np
import torch

from oracle import immoco_oracle as orc


def test_level_geometry_2d_3d():
    g2 = orc.geometry_from_config(2, orc.encoding_config)
    assert g2.resolutions[:7] == [16, 32, 64, 128, 256, 512, 1024]
    assert g2.sizes[:6] == [256, 1024, 4096, 16384, 65536, 262144] and set(g2.sizes[6:]) == {524288}
    assert g2.offsets[:7] == [0, 256, 1280, 5376, 21760, 87296, 349440]
    # levels 12-15 (res >= 2^16): upstream's uint32 `stride *= resolution` wraps to 0, so `hashmap_size < stride`
    # is false and the level keeps the dense (wrapped) index - see test_wrapped_stride_levels_known_answers
    assert g2.hashed == [False] * 6 + [True] * 6 + [False] * 4
    assert g2.n_entries == 5592320
    g3 = orc.geometry_from_config(3, orc.encoding_config)
    assert g3.sizes[:3] == [4096, 32768, 262144] and g3.offsets[:4] == [0, 4096, 36864, 299008]
    assert g3.hashed == [False] * 3 + [True] * 9 + [False] * 4
    assert g3.n_entries == 7114752
    assert [s for s in g3.scales[:3]] == [15.0, 31.0, 63.0]
    # parameter counts quoted in SURVEY a5/a6
    m_img = orc.mlp_spec_from_config(32, 2, orc.network_config)
    m_mot = orc.mlp_spec_from_config(32, 2, orc.mot_network_config)
    assert m_img.n_params + g2.n_table_params == 11194880
    assert m_mot.n_params + g3.n_table_params == 14232576


def test_hash_known_answers():
    g3 = orc.geometry_from_config(3, orc.encoding_config)
    cell = np.array([[1, 2, 3], [0, 0, 0], [0xFFFFFFFF, 5, 7]], dtype=np.uint64)
    lvl = 5
    assert g3.hashed[lvl]
    exp = []
    for c in cell.tolist():
        h = 0
        for d in range(3):
            h ^= (c[d] * orc.PRIMES[d]) & 0xFFFFFFFF
        exp.append(h % 524288)
    assert orc.grid_index(cell, g3, lvl).tolist() == exp
    assert exp[0] == ((1 ^ ((2 * 2654435761) & 0xFFFFFFFF) ^ ((3 * 805459861) & 0xFFFFFFFF)) % 524288)


def test_wrapped_stride_levels_known_answers():
    """tiny-cuda-nn grid_index(): `uint32_t stride = 1; for (dim < N_DIMS && stride <= hashmap_size)
    { index += pos[dim] * stride; stride *= resolution; } if (hashmap_size < stride) index = hash(pos);`
    With res = 2^16 .. 2^19 (levels 12-15 of base 16 / scale 2) the second multiplication wraps to exactly 0:
    no hash, index = (c0 + c1 * res) mod 2^32 mod 2^19, and a third dimension is multiplied by stride 0."""
    T = 1 << 19
    for dims in (2, 3):
        geo = orc.geometry_from_config(dims, orc.encoding_config)
        for lvl in (12, 13, 14, 15):
            res = geo.resolutions[lvl]
            assert res == 1 << (4 + lvl) and geo.sizes[lvl] == T and not geo.hashed[lvl]
            # the same walk in plain Python integers
            stride, walked = 1, []
            for d in range(dims):
                if stride > T:
                    break
                walked.append(stride)
                stride = (stride * res) & 0xFFFFFFFF
            assert stride == 0 and walked == [1, res] + [0] * (dims - 2)
            cells = np.array([[1, 2, 3], [0xFFFFFFF1, 7, 99], [65535, 0xFFFFFFFF, 5], [2, 9, 0]], dtype=np.uint64)[:, :dims]
            got = orc.grid_index(cells, geo, lvl).tolist()
            exp = [int((int(c[0]) + int(c[1]) * res) & 0xFFFFFFFF) % T for c in cells]
            assert got == exp
        # level 15 (res = 2^19): c1 * 2^19 vanishes mod 2^19 -> the index is c0 alone
        c = np.array([[5, 123456, 77][:dims], [5, 1, 2][:dims]], dtype=np.uint64)
        assert orc.grid_index(c, geo, 15).tolist() == [5, 5]
        # level 12 (res = 2^16): only the low 3 bits of c1 survive
        c = np.array([[5, 8 + 3, 1][:dims], [5, 3, 2][:dims]], dtype=np.uint64)
        assert orc.grid_index(c, geo, 12).tolist() == [5 + 3 * 65536] * 2
        # level 11 (res = 2^15): res^2 = 2^30 > 2^19, no wrap -> still hashed
        assert geo.hashed[11]
    # a non-power-of-two scale does not wrap to 0: hashing stays on the fine levels
    geo = orc.geometry_from_config(2, dict(orc.encoding_config, per_level_scale=1.9))
    assert all(geo.hashed[l] for l in range(8, 16))


def test_dense_index_negative_wrap():
    """x in [-1,0) gives negative cells that wrap mod 2^32."""
    g2 = orc.geometry_from_config(2, orc.encoding_config)
    coords = np.array([[-1.0, -1.0], [1.0, 1.0], [-0.5, 0.25]], dtype=np.float32)
    cell, frac = orc.grid_cells(coords, g2, 0)          # scale 15: pos = 15x+0.5
    assert cell[0].tolist() == [0xFFFFFFF1, 0xFFFFFFF1] and np.allclose(frac[0], 0.5)
    assert cell[1].tolist() == [15, 15]
    assert cell[2].tolist() == [0xFFFFFFF9, 4]            # floor(-7.0)=-7 ; floor(4.25)=4
    idx = orc.grid_index(cell, g2, 0)
    assert idx[0] == ((0xFFFFFFF1 + 0xFFFFFFF1 * 16) & 0xFFFFFFFF) % 256
    assert idx[1] == (15 + 15 * 16) % 256


def test_interpolation_weights_and_linearity():
    g = torch.Generator().manual_seed(0)
    coords = (torch.rand(257, 3, generator=g) * 2 - 1)
    g3 = orc.geometry_from_config(3, orc.encoding_config)
    plan = orc.HashGridPlan(coords, g3)
    assert plan.idx.shape == (257, 16, 8)
    np.testing.assert_allclose(plan.w.sum(-1).numpy(), 1.0, atol=1e-6)
    for l in range(16):
        assert int(plan.idx[:, l].min()) >= g3.offsets[l] and int(plan.idx[:, l].max()) < g3.offsets[l + 1]
    # a constant table encodes to that constant
    tab = torch.ones(g3.n_entries, 2) * torch.tensor([0.25, -2.0])
    enc = plan.encode(tab).view(257, 16, 2)
    np.testing.assert_allclose(enc[..., 0].numpy(), 0.25, rtol=1e-6)
    np.testing.assert_allclose(enc[..., 1].numpy(), -2.0, rtol=1e-6)


def test_inr_gradient_vs_float64_finite_difference():
    """fp32 autograd gradient of the oracle INR == float64 central differences."""
    torch.manual_seed(0)
    inr = orc.OracleINR(2, 2, orc.encoding_config, orc.network_config, seed=5)
    x = orc.make_grids((6, 5))
    tgt = torch.randn(30, 2)
    plan = inr.plan_for(x)

    def f64(p):
        w1, w2, tab = inr.split(p)
        h = torch.relu(plan.encode(tab) @ w1.t())
        return (((h @ w2.t())[:, :2] - tgt.double()) ** 2).sum()

    loss = ((inr(x) - tgt) ** 2).sum()
    loss.backward()
    g = inr.params.grad.clone()
    p64 = inr.params.detach().double()
    idx = torch.cat([torch.arange(0, 4), torch.arange(inr.mlp.n_w1, inr.mlp.n_w1 + 4),
                     g[inr.mlp.n_params:].abs().topk(6).indices + inr.mlp.n_params])
    for i in idx.tolist():
        eps = 1e-7
        pp, pm = p64.clone(), p64.clone()
        pp[i] += eps
        pm[i] -= eps
        fd = float((f64(pp) - f64(pm)) / (2 * eps))
        assert abs(fd - g[i].item()) <= 1e-2 * abs(fd) + 1e-7, (i, fd, g[i].item())


def test_init_distribution():
    g2 = orc.geometry_from_config(2, orc.encoding_config)
    m = orc.mlp_spec_from_config(32, 2, orc.network_config)
    p = orc.init_inr_params(g2, m, 1337)
    assert p.shape == (11194880,) and p.dtype == np.float32
    b1 = np.sqrt(6 / (32 + 256))
    assert np.abs(p[: m.n_w1]).max() <= b1 and np.abs(p[: m.n_w1]).max() > 0.98 * b1
    t = p[m.n_params:]
    assert np.abs(t).max() <= 1e-4 and abs(t.mean()) < 1e-7 and abs(t.std() - 1e-4 / np.sqrt(3)) < 1e-7
    # different seeds / streams decorrelate
    q = orc.init_inr_params(g2, m, 1338)
    assert abs(np.corrcoef(p[:8192], q[:8192])[0, 1]) < 0.05


def test_c_encode_matches_torch_expression():
    """oracle/hashgrid_oracle.c (the C evaluation of the interpolation) vs the torch expression it restates:
    forward bit-exact, backward to fp32 rounding, every summation order."""
    g = torch.Generator().manual_seed(3)
    for dims, n in ((2, 9001), (3, 5003)):
        geo = orc.geometry_from_config(dims, orc.encoding_config)
        coords = torch.rand(n, dims, generator=g) * 2 - 1
        table = torch.rand(geo.n_entries, 2, generator=g) - 0.5
        denc = torch.randn(n, 32, generator=g)
        plan = orc.HashGridPlan(coords, geo)
        t0 = table.clone().requires_grad_(True)
        e0 = plan.encode_torch(t0)
        (e0 * denc).sum().backward()
        for order in (0, 1, 2, 7):
            plan.bwd_order = order
            t1 = table.clone().requires_grad_(True)
            e1 = plan.encode_c(t1)
            assert torch.equal(e1, e0)
            (e1 * denc).sum().backward()
            assert (t1.grad - t0.grad).abs().max() <= 1e-5 * t0.grad.abs().max()
            assert torch.equal(t1.grad == 0, t0.grad == 0)      # the same entries are touched
    # lattice with many points per entry (the wrapped-stride levels): large sums stay within rounding
    geo = orc.geometry_from_config(3, orc.encoding_config)
    x = orc.make_grids((2, 40, 40))
    plan = orc.HashGridPlan(x, geo)
    denc = torch.randn(x.shape[0], 32, generator=g)
    t0 = torch.zeros(geo.n_entries, 2, requires_grad=True)
    (plan.encode_torch(t0) * denc).sum().backward()
    t1 = torch.zeros(geo.n_entries, 2, requires_grad=True)
    (plan.encode_c(t1) * denc).sum().backward()
    assert (t1.grad - t0.grad).abs().max() <= 2e-5 * t0.grad.abs().max()


def test_redraw_mode_is_the_same_function_in_another_summation_order():
    """OracleIMMoCo.redraw (round 3: new fp32 summation orders before EVERY step - hash-grid backward block order, MLP
    batch row order, motion-group order; the per-step analogue of tiny-cuda-nn's nondeterministic atomics): the
    forward is bit-identical, the gradients agree to summation accuracy and are NOT bit-identical."""
    import torch.nn.functional as F
    from oracle import synth_cpu
    s = synth_cpu.make_slice(64, 64, 3, 5)
    masks = orc.extract_movement_groups(s["lines"], make_list=True)
    model = orc.OracleIMMoCo(masks)
    k = s["kspace"]
    kin = k.div(k.abs().max()).mul(16000)

    def grads():
        model.zero_grad()
        kf, ip = model()
        loss = F.mse_loss(torch.view_as_real(kf), torch.view_as_real(kin)) + orc.gradient_entropy_loss(ip) * 1e-2
        loss.backward()
        return float(loss), ip.detach().clone(), model.image_inr.params.grad.clone(), model.motion_inr.params.grad.clone()

    l0, i0, gi0, gm0 = grads()
    model.redraw(np.random.default_rng(3))
    l1, i1, gi1, gm1 = grads()
    assert torch.equal(i0, i1) and abs(l0 - l1) <= 1e-6 * abs(l0)
    for a, b in ((gi0, gi1), (gm0, gm1)):
        rel = float((a - b).norm() / a.norm())
        assert 0.0 < rel <= 1e-5, rel


def test_mlp_fp16_mode_rounds_what_tcnn_rounds():
    """OracleINR(mlp_fp16=True) (_MLPHalf): fp16 operands, fp32 accumulation, loss scale 128; with denc_fp16 the
    scaled dL/denc is rounded to fp16 as well.  Close to, not equal to, the fp32 network; the loss scale itself
    changes nothing but the rounding (a power of two)."""
    torch.manual_seed(0)
    x = torch.rand(700, 3) * 2 - 1
    kw = dict(seed=3)
    a = orc.OracleINR(3, 2, orc.encoding_config, orc.mot_network_config, **kw)
    b = orc.OracleINR(3, 2, orc.encoding_config, orc.mot_network_config, mlp_fp16=True, **kw)
    c = orc.OracleINR(3, 2, orc.encoding_config, orc.mot_network_config, mlp_fp16=True, denc_fp16=False, **kw)
    with torch.no_grad():
        a.params[a.mlp.n_params:] *= 1e3           # features of order 0.1 instead of 1e-4
        b.params.copy_(a.params)
        c.params.copy_(a.params)
    g = torch.randn(700, 2)
    outs = []
    for m in (a, b, c):
        o = m(x)
        o.backward(g)
        outs.append((o.detach(), m.params.grad.clone()))
    assert torch.equal(outs[1][0], outs[2][0])                         # same forward
    d_out = float((outs[0][0] - outs[1][0]).norm() / outs[0][0].norm())
    d_g = float((outs[0][1] - outs[1][1]).norm() / outs[0][1].norm())
    assert 1e-5 < d_out < 5e-3 and 1e-5 < d_g < 5e-3, (d_out, d_g)
    nw = a.mlp.n_params
    assert torch.equal(outs[1][1][:nw], outs[2][1][:nw])              # weight gradients do not see the denc rounding
    d_t = float((outs[1][1][nw:] - outs[2][1][nw:]).norm() / outs[2][1][nw:].norm())
    assert 0.0 < d_t < 2e-3, d_t                                       # table gradients do (2^-11 per contribution)


def test_device_encode_expression_is_the_torch_expression():
    """HashGridPlan.encode_device (the device oracle's encode: one index_select per corner, backward = index_add_) is
    the same arithmetic as encode_torch; checked here on CPU tensors (forward bit-identical, gradient to summation
    accuracy), on the device by the teacher-forced tests of tests/test_gpu_ops.py."""
    geo = orc.geometry_from_config(3, orc.encoding_config)
    x = orc.make_grids((3, 12, 10))
    plan = orc.HashGridPlan(x, geo)
    g = torch.Generator().manual_seed(5)
    t0 = torch.randn(geo.n_entries, 2, generator=g).requires_grad_(True)
    t1 = t0.detach().clone().requires_grad_(True)
    e0, e1 = plan.encode_torch(t0), plan.encode_device(t1)
    assert torch.equal(e0, e1)
    d = torch.randn(e0.shape, generator=g)
    (e0 * d).sum().backward()
    (e1 * d).sum().backward()
    assert (t0.grad - t1.grad).abs().max() <= 2e-5 * t0.grad.abs().max()
    # default seeds: both INRs start from tiny-cuda-nn's module default 1337 (immoco.py:60-65)
    m = orc.OracleIMMoCo(torch.zeros(1, 8, 8, dtype=torch.long))
    assert m.image_inr.seed == m.motion_inr.seed == 1337
