"""T2 kernel correctness: every non-conv HIP kernel against the plain-PyTorch fp32 reference of the same op
(ops/ref_ops.py run on CPU copies of the same inputs)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]


def _tol(dtype):
    return (1e-4, 1e-5) if dtype == torch.float32 else (2e-2, 2e-2)


def _rand(shape, dtype, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(4, 14, 14, 64), (2, 7, 9, 24), (3, 5, 5, 2048)])
def test_bn_forward_backward(shape, dtype):
    from distributedpytorch_amd.ops import hip_ops as H, ref_ops as R
    c = shape[3]
    m = shape[0] * shape[1] * shape[2]
    x = _rand(shape, dtype, 1, 2.0) + 0.5
    res = _rand(shape, dtype, 2)
    gamma, beta = torch.rand(c) + 0.5, torch.randn(c)
    outs = {}
    for name, O, dev in (("ref", R, "cpu"), ("hip", H, "cuda")):
        xd, rd = x.to(dev), res.to(dev)
        g, b = gamma.to(dev), beta.to(dev)
        rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        nbt = torch.zeros(1, dtype=torch.int64, device=dev)
        rows = O.bn_slab_rows(m)
        slab = torch.zeros(rows, 2, c, device=dev)
        O.bn_stats(xd, slab)
        mean, invstd, scale, shift = (torch.zeros(c, device=dev) for _ in range(4))
        O.bn_finalize(slab, m, g, b, rm, rv, nbt, 0.1, 1e-5, mean, invstd, scale, shift)
        y = torch.empty_like(xd)
        O.bn_apply(xd, scale, shift, y, residual=rd, relu=True)
        dy = _rand(shape, dtype, 3).to(dev)
        dg, db = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
        sums = torch.zeros(2, c, device=dev)
        coef = torch.zeros(3, c, device=dev)
        O.bn_bwd_reduce(dy, y, xd, mean, invstd, g, dg, db, sums, coef)
        dx = torch.empty_like(xd)
        O.bn_bwd_dx(dy, xd, coef, dx)
        outs[name] = [t.float().cpu() for t in (mean, invstd, rm, rv, y, dy, dg, db, dx)] + [nbt.cpu()]
    rtol, atol = _tol(dtype)
    names = "mean invstd rm rv y dz dgamma dbeta dx nbt".split()
    for n, a, b in zip(names, outs["hip"], outs["ref"]):
        sc = max(1.0, float(b.abs().max())) if n in ("dgamma", "dbeta", "dx") else 1.0
        torch.testing.assert_close(a, b, rtol=rtol * 5, atol=atol * 5 * sc, msg=lambda s, n=n: f"{n}: {s}")


@pytest.mark.parametrize("dtype", DTYPES)
def test_bn_backward_mask_recomputed_from_x(dtype):
    """BN+ReLU without a residual: the ReLU mask recomputed from x with the forward's folded coefficients must give the
    same dgamma / dbeta / dx as the stored-output mask, and must leave dy untouched."""
    from distributedpytorch_amd.ops import hip_ops as H
    shape = (4, 9, 11, 72)
    c, m = shape[3], shape[0] * shape[1] * shape[2]
    x = (_rand(shape, dtype, 1, 2.0) + 0.3).cuda()
    g, b = (torch.rand(c) + 0.5).cuda(), torch.randn(c).cuda()
    rm, rv, nbt = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda"), torch.zeros(1, dtype=torch.int64, device="cuda")
    slab = torch.zeros(H.bn_slab_rows(m), 2, c, device="cuda")
    H.bn_stats(x, slab)
    mean, invstd, scale, shift = (torch.zeros(c, device="cuda") for _ in range(4))
    H.bn_finalize(slab, m, g, b, rm, rv, nbt, 0.1, 1e-5, mean, invstd, scale, shift)
    y = torch.empty_like(x)
    H.bn_apply(x, scale, shift, y, relu=True)
    dy0 = _rand(shape, dtype, 3).cuda()
    res = []
    for from_x in (False, True):
        dy = dy0.clone()
        dg, db = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda")
        sums, coef = torch.zeros(2, c, device="cuda"), torch.zeros(3, c, device="cuda")
        kw = dict(mask_scale=scale, mask_shift=shift) if from_x else {}
        H.bn_bwd_reduce(dy, None if from_x else y, x, mean, invstd, g, dg, db, sums, coef, **kw)
        dx = torch.empty_like(x)
        H.bn_bwd_dx(dy, x, coef, dx, **kw)
        res.append((dg.clone(), db.clone(), dx.float().clone(), dy.clone()))
    assert torch.equal(res[1][3], dy0)                       # dy untouched in the recompute mode
    assert not torch.equal(res[0][3], dy0)                   # masked in place in the stored-output mode
    tol = dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    for a_, b_ in zip(res[0][:3], res[1][:3]):
        torch.testing.assert_close(a_, b_, **tol)


@pytest.mark.parametrize("dtype", DTYPES)
def test_bn_eval_and_channel_slices(dtype):
    from distributedpytorch_amd.ops import hip_ops as H, ref_ops as R
    buf = _rand((2, 6, 6, 64), dtype, 5)
    c = 32
    gamma, beta, rm, rv = torch.rand(c) + 0.5, torch.randn(c), torch.randn(c), torch.rand(c) + 0.5
    res = {}
    for name, O, dev in (("ref", R, "cpu"), ("hip", H, "cuda")):
        b = buf.to(dev)
        x = b[..., 16:48]
        scale, shift = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
        O.bn_eval_coeffs(gamma.to(dev), beta.to(dev), rm.to(dev), rv.to(dev), 1e-5, scale, shift)
        outb = torch.zeros(2, 6, 6, 96, device=dev, dtype=dtype)
        O.bn_apply(x, scale, shift, outb[..., 64:96], relu=False)
        res[name] = outb.float().cpu()
    rtol, atol = _tol(dtype)
    torch.testing.assert_close(res["hip"], res["ref"], rtol=rtol, atol=atol)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [((2, 112, 112, 64), 3, 2, 1, False), ((2, 55, 55, 64), 3, 2, 0, False),
                                 ((2, 109, 109, 96), 3, 2, 0, True), ((2, 14, 14, 512), 2, 2, 0, False)])
def test_maxpool(cfg, dtype):
    from distributedpytorch_amd.ops import hip_ops as H, ref_ops as R
    shape, k, s, p, ceil = cfg
    n, h, w, c = shape
    ho, wo = R.pool_out_size(h, k, s, p, ceil), R.pool_out_size(w, k, s, p, ceil)
    x = _rand(shape, dtype, 1).clamp_min(0)     # post-ReLU-like input with ties at zero
    dy = _rand((n, ho, wo, c), dtype, 2)
    res = {}
    for name, O, dev in (("ref", R, "cpu"), ("hip", H, "cuda")):
        xd = x.to(dev)
        y = torch.empty(n, ho, wo, c, device=dev, dtype=dtype)
        idx = torch.empty(n, ho, wo, c, device=dev, dtype=torch.uint8)
        O.maxpool_fwd(xd, y, idx, (k, k), (s, s), (p, p))
        dx = torch.empty_like(xd)
        O.maxpool_bwd(dy.to(dev), idx, dx, (k, k), (s, s), (p, p))
        res[name] = (y.float().cpu(), dx.float().cpu())
    torch.testing.assert_close(res["hip"][0], res["ref"][0], rtol=0, atol=0)
    ref_nchw = torch.nn.functional.max_pool2d(x.float().permute(0, 3, 1, 2), k, s, p, ceil_mode=ceil).permute(0, 2, 3, 1)
    torch.testing.assert_close(res["hip"][0], ref_nchw, rtol=0, atol=0)
    # backward, per element (the post-ReLU input above has ties at zero, where the winner inside a window is a convention:
    # there only the gradient MASS per channel can be compared; the element-wise check uses a tie-free fp32 input and
    # torch's own max-pool backward - the same kind of check as tests/test_gpu_parity.py::test_maxpool_backward_per_element)
    rtol, atol = _tol(dtype)
    torch.testing.assert_close(res["hip"][1].sum((1, 2)), res["ref"][1].sum((1, 2)), rtol=rtol * 10, atol=atol * 50)
    if dtype == torch.float32:
        x2 = _rand(shape, dtype, 3)                 # continuous values: no two equal in a window
        xt = x2.permute(0, 3, 1, 2).clone().requires_grad_(True)
        yt = torch.nn.functional.max_pool2d(xt, k, s, p, ceil_mode=ceil)
        yt.backward(dy.float().permute(0, 3, 1, 2))
        y = torch.empty(n, ho, wo, c, device="cuda", dtype=dtype)
        idx = torch.empty(n, ho, wo, c, device="cuda", dtype=torch.uint8)
        H.maxpool_fwd(x2.cuda(), y, idx, (k, k), (s, s), (p, p))
        dx = torch.empty(shape, device="cuda", dtype=dtype)
        H.maxpool_bwd(dy.cuda(), idx, dx, (k, k), (s, s), (p, p))
        torch.testing.assert_close(dx.cpu(), xt.grad.permute(0, 2, 3, 1), rtol=1e-6, atol=1e-6)      # every element, not the sum


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [((2, 7, 7, 512), None, None, 0, (1, 1)), ((2, 13, 13, 256), None, None, 0, (6, 6)),
                                 ((2, 14, 14, 512), None, None, 0, (7, 7)), ((2, 28, 28, 128), 2, 2, 0, None),
                                 ((2, 17, 17, 768), 5, 3, 0, None), ((2, 35, 35, 192), 3, 1, 1, None)])
def test_avgpool(cfg, dtype):
    from distributedpytorch_amd.ops import hip_ops as H, ref_ops as R
    shape, k, s, p, adaptive = cfg
    n, h, w, c = shape
    if adaptive:
        ho, wo = adaptive
        ks = st = None
    else:
        ho, wo = R.pool_out_size(h, k, s, p), R.pool_out_size(w, k, s, p)
        ks, st = (k, k), (s, s)
    x = _rand(shape, dtype, 1)
    dy = _rand((n, ho, wo, c), dtype, 2)
    res = {}
    for name, O, dev in (("ref", R, "cpu"), ("hip", H, "cuda")):
        xd = x.to(dev)
        y = torch.empty(n, ho, wo, c, device=dev, dtype=dtype)
        O.avgpool_fwd(xd, y, ks, st, (p, p), adaptive=bool(adaptive))
        dx = torch.empty_like(xd)
        O.avgpool_bwd(dy.to(dev), dx, ks, st, (p, p), adaptive=bool(adaptive))
        res[name] = (y.float().cpu(), dx.float().cpu())
    rtol, atol = _tol(dtype)
    torch.testing.assert_close(res["hip"][0], res["ref"][0], rtol=rtol, atol=atol)
    torch.testing.assert_close(res["hip"][1], res["ref"][1], rtol=rtol, atol=atol)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("nc", [(64, 10), (37, 1000), (203, 1000)])     # the last one takes the multi-block path
def test_ce_loss(nc, mode, dtype):
    from distributedpytorch_amd.ops import hip_ops as H, ref_ops as R
    n, c = nc
    ld = (c + 7) // 8 * 8
    z = _rand((n, ld), dtype, 1, 3.0)
    labels = torch.randint(0, c, (n,), generator=torch.Generator().manual_seed(3))
    cw = torch.rand(c) + 0.5
    res = {}
    for name, O, dev in (("ref", R, "cpu"), ("hip", H, "cuda")):
        out = torch.zeros(2, device=dev)
        accum = torch.zeros(3, device=dev)
        dl = torch.full((n, ld), 7.0, device=dev, dtype=dtype)
        O.ce_loss(z.to(dev), labels.to(dev), c, out, accum=accum, dlogits=dl, class_w=cw.to(dev) if mode else None,
                  mode=mode, scale=1.0)
        O.ce_loss(z.to(dev), labels.to(dev), c, out, accum=accum, dlogits=None, class_w=cw.to(dev) if mode else None,
                  mode=mode, scale=0.4, aux=True)
        res[name] = (out.cpu(), accum.cpu(), dl.float().cpu())
    # oracle for the loss value itself
    zf = z[:, :c].float()
    if mode == 0:
        expect = torch.nn.functional.cross_entropy(zf, labels)
    elif mode == 1:
        expect = torch.nn.functional.cross_entropy(zf, labels, weight=cw)
    else:
        lp = torch.log_softmax(zf, -1)
        expect = torch.nn.functional.nll_loss(((1 - lp.exp()) ** 2) * lp, labels, weight=cw, reduction="none").mean()
    torch.testing.assert_close(res["ref"][0][0], expect * 1.4, rtol=1e-4, atol=1e-5)
    rtol, atol = _tol(dtype)
    for a, b in zip(res["hip"], res["ref"]):
        torch.testing.assert_close(a, b, rtol=max(rtol, 1e-3), atol=atol)


@pytest.mark.parametrize("dtype", DTYPES)
def test_elementwise_and_colsum(dtype):
    from distributedpytorch_amd.ops import hip_ops as H, ref_ops as R
    a, b = _rand((3, 9, 9, 40), dtype, 1), _rand((3, 9, 9, 40), dtype, 2)
    res = {}
    for name, O, dev in (("ref", R, "cpu"), ("hip", H, "cuda")):
        ad, bd = a.to(dev), b.to(dev)
        s = torch.empty_like(ad)
        O.add(ad, bd, s)
        r = torch.empty_like(ad)
        O.relu_fwd(ad, r)
        g = bd.clone()
        O.relu_bwd(g, r)
        cs = torch.ones(40, device=dev)
        O.colsum(ad, cs, accumulate=True)
        big = _rand((4, 40, 40, 16), dtype, 4).to(dev)
        cs2 = torch.zeros(16, device=dev)
        O.colsum(big, cs2)
        # ReLU backward + bias gradient in one pass (tall: the fused kernel; short: the two separate ones), sliced views
        yb = _rand((4, 40, 40, 24), dtype, 6).to(dev)
        gb = _rand((4, 40, 40, 24), dtype, 7).to(dev)
        cs3 = torch.ones(16, device=dev)
        O.relu_bwd_colsum(gb[..., 8:], yb[..., 8:], cs3, accumulate=True)
        ys, gs = _rand((2, 5, 5, 16), dtype, 8).to(dev), _rand((2, 5, 5, 16), dtype, 9).to(dev)
        cs4 = torch.zeros(16, device=dev)
        O.relu_bwd_colsum(gs, ys, cs4)
        w = _rand((24, 3, 3, 16), dtype, 5).to(dev)
        wt = torch.empty(16, 3, 3, 24, device=dev, dtype=dtype)
        O.transpose_krsc(w, wt)
        res[name] = [t.float().cpu() for t in (s, r, g, cs, cs2, wt, gb, cs3, gs, cs4)]
    rtol, atol = _tol(dtype)
    for x, y in zip(res["hip"], res["ref"]):
        torch.testing.assert_close(x, y, rtol=rtol, atol=atol * 20)


@pytest.mark.parametrize("dtype", DTYPES)
def test_dropout_mask_regenerates(dtype):
    from distributedpytorch_amd.ops import hip_ops as H
    x = torch.ones(8, 1, 1, 4096, device="cuda", dtype=dtype)
    rng = torch.tensor([1234, 0], dtype=torch.int64, device="cuda")
    y1, y2 = torch.empty_like(x), torch.empty_like(x)
    H.dropout(x, y1, 0.5, rng, 3)
    H.dropout(x, y2, 0.5, rng, 3)
    assert torch.equal(y1, y2)                          # backward regenerates the same mask
    keep = (y1 != 0).float().mean().item()
    assert 0.47 < keep < 0.53
    assert set(y1.float().unique().tolist()) <= {0.0, 2.0}
    H.rng_advance(rng)
    H.dropout(x, y2, 0.5, rng, 3)
    assert not torch.equal(y1, y2)                      # next step, new mask
    y3 = torch.empty_like(x)
    H.dropout(x, y3, 0.5, rng, 4)
    assert not torch.equal(y2, y3)                      # other layer, other mask


def test_optimizers_match_torch():
    """Fused arena Adam / SGD-momentum against torch.optim, step by step FROM THE SAME STATE (parameters and moments are
    re-synchronised to torch's before every step, so each step's own arithmetic is what is compared): within 4 fp32 ulps
    at the scale of the parameter or of its update, whichever is larger.  optim.hip is built with IEEE arithmetic
    (no -ffast-math since round 3) and reproduces torch's scalar handling (double betas, float(1 - beta) increments,
    lerp form of the first moment); what is left is the order of alpha * m / denom in torch's CPU addcdiv."""
    from distributedpytorch_amd.ops import hip_ops as H
    from distributedpytorch_amd.nn.optim import make_hyper, STEP
    n = 10_007
    for kind in ("adam", "sgd"):
        p0, g0 = torch.randn(n), torch.randn(n)
        ref = torch.nn.Parameter(p0.clone())
        opt = torch.optim.Adam([ref], lr=1e-3) if kind == "adam" else torch.optim.SGD([ref], lr=1e-3, momentum=0.9)
        hyper = make_hyper(1e-3, .9, .999, 1e-8, 0., 0., 0.9, 0.5, device="cuda")
        p16 = torch.zeros(n, device="cuda", dtype=torch.bfloat16)
        worst = 0.0
        for it in range(6):
            st = opt.state.get(ref, {})
            p = ref.data.clone().cuda()
            if kind == "adam":
                m = st["exp_avg"].clone().cuda() if st else torch.zeros(n, device="cuda")
                v = st["exp_avg_sq"].clone().cuda() if st else torch.zeros(n, device="cuda")
            else:
                m = st["momentum_buffer"].clone().cuda() if st else torch.zeros(n, device="cuda")
            hyper[STEP] = float(it)
            g = (g0 * (it + 1)).cuda()
            if kind == "adam":
                H.adam_step(p, g * 2, m, v, p16, hyper)          # grad_scale 0.5 folds the DDP 1/world
            else:
                H.sgd_step(p, g * 2, m, p16, hyper)
            ref.grad = g0 * (it + 1)
            before = ref.data.clone()
            opt.step()
            scale = torch.maximum(ref.data.abs(), (ref.data - before).abs())      # the parameter or its update, whichever is larger
            ulp = torch.exp2(torch.floor(torch.log2(scale)) - 23)
            worst = max(worst, ((p.cpu() - ref.data).abs() / ulp).max().item())
            st = opt.state[ref]
            if kind == "adam":
                torch.testing.assert_close(m.cpu(), st["exp_avg"], rtol=3e-7, atol=1e-12)
                torch.testing.assert_close(v.cpu(), st["exp_avg_sq"], rtol=3e-7, atol=1e-12)
            else:
                torch.testing.assert_close(m.cpu(), st["momentum_buffer"], rtol=3e-7, atol=1e-12)
            torch.testing.assert_close(p16.float().cpu(), ref.data.bfloat16().float(), rtol=1e-2, atol=1e-2)
        # measured: adam 3.0, sgd <= 2 (torch's CPU addcdiv rounds (alpha * m) / denom, its GPU kernel and ours alpha * (m / denom);
        # bias-correction casts and sqrt add an ulp).  Before round 3 (-ffast-math, float-derived betas): 27.5.
        assert worst <= 4.0, (kind, worst)


@pytest.mark.parametrize("dtype", DTYPES)
def test_augment_mnist(dtype):
    from distributedpytorch_amd.ops import hip_ops as H
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(0)
    images = torch.randint(0, 256, (50, 28, 28), generator=g, dtype=torch.uint8)
    labels = torch.randint(0, 10, (50,), generator=g)
    idx = torch.tensor([3, 7, 49, 0], dtype=torch.int32)
    cpad = 8
    out = torch.full((4, 224, 224, cpad), float("nan"), device="cuda", dtype=dtype)
    lab = torch.zeros(4, dtype=torch.int64, device="cuda")
    # eval transform == bilinear resize + normalise + 3-channel repeat: exact oracle
    H.augment_mnist(images.cuda(), labels.cuda(), idx.cuda(), out, lab, False, 0.1307, 0.3081, 1234, 0)
    ref = F.interpolate(images[idx.long()].float()[:, None], size=(224, 224), mode="bilinear", align_corners=False)[:, 0]
    ref = (ref / 255 - 0.1307) / 0.3081
    rtol, atol = _tol(dtype)
    for ch in range(3):
        torch.testing.assert_close(out[..., ch].float().cpu(), ref, rtol=rtol, atol=atol * 10)
    assert out[..., 3:].abs().max().item() == 0
    assert torch.equal(lab.cpu(), labels[idx.long()])
    # train transform: deterministic in (seed, epoch, index), changes with the epoch, stays in range
    o1, o2, o3 = (torch.empty_like(out) for _ in range(3))
    H.augment_mnist(images.cuda(), labels.cuda(), idx.cuda(), o1, lab, True, 0.1307, 0.3081, 1234, 0)
    H.augment_mnist(images.cuda(), labels.cuda(), idx.cuda(), o2, lab, True, 0.1307, 0.3081, 1234, 0)
    H.augment_mnist(images.cuda(), labels.cuda(), idx.cuda(), o3, lab, True, 0.1307, 0.3081, 1234, 1)
    assert torch.equal(o1, o2) and not torch.equal(o1, o3)
    lo, hi = (0 - 0.1307) / 0.3081, (1 - 0.1307) / 0.3081
    assert o1[..., 0].float().min().item() >= lo - 0.05 and o1[..., 0].float().max().item() <= hi + 0.05


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(4, 28, 28, 64, 3, 2, 1), (2, 17, 13, 32, 3, 2, 1), (3, 12, 12, 16, 2, 2, 0)])
def test_fused_bn_relu_maxpool_forward(shape, dtype):
    """maxpool_fwd(scale=, shift=) == maxpool(relu(x * scale + shift)), values and argmax taps, bit for bit."""
    from distributedpytorch_amd.ops import hip_ops as H
    n, h, w, c, k, s, pd = shape
    g = torch.Generator(device="cuda").manual_seed(5)
    z = torch.randn(n, h, w, c, device="cuda", generator=g).to(dtype)
    sc = torch.rand(c, device="cuda", generator=g) + 0.5
    sh = torch.randn(c, device="cuda", generator=g) * 0.3
    mean = torch.randn(c, device="cuda", generator=g) * 0.1
    invstd = torch.rand(c, device="cuda", generator=g) + 0.5
    gamma = torch.rand(c, device="cuda", generator=g) + 0.5
    ho, wo = H.pool_out_size(h, k, s, pd), H.pool_out_size(w, k, s, pd)
    # forward
    act = torch.empty_like(z)
    H.bn_apply(z, sc, sh, act, relu=True)
    ref_y = torch.empty(n, ho, wo, c, device="cuda", dtype=dtype)
    ref_i = torch.empty(n, ho, wo, c, device="cuda", dtype=torch.uint8)
    H.maxpool_fwd(act, ref_y, ref_i, (k, k), (s, s), (pd, pd))
    y = torch.empty_like(ref_y)
    idx = torch.empty_like(ref_i)
    H.maxpool_fwd(z, y, idx, (k, k), (s, s), (pd, pd), scale=sc, shift=sh)
    assert torch.equal(y, ref_y) and torch.equal(idx, ref_i)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("m,c", [(1000, 64), (4099, 256), (77, 1024), (513, 2048), (300, 32)])
def test_bn_apply_writes_the_relu_mask_as_bits(m, c, dtype):
    """bn_apply(mask_bits=): bit b of word w of a row = (y[row, 32 w + b] > 0), for every channel-chunk / row-lane shape of
    the kernel (a thread owns 4 or 8 channels; the lanes of a word combine their bits with shuffles)."""
    from distributedpytorch_amd.ops import hip_ops
    g = torch.Generator(device="cuda").manual_seed(m + c)
    x = torch.randn(m, 1, 1, c, device="cuda", generator=g).to(dtype)
    res = torch.randn(m, 1, 1, c, device="cuda", generator=g).to(dtype)
    scale = torch.rand(c, device="cuda", generator=g) + 0.5
    shift = torch.randn(c, device="cuda", generator=g) * 0.3
    for residual in (res, None):
        y = torch.empty_like(x)
        bits = torch.full((m, c // 32), -1, device="cuda", dtype=torch.int32)
        hip_ops.bn_apply(x, scale, shift, y, residual=residual, relu=True, mask_bits=bits)
        y2 = torch.empty_like(x)
        hip_ops.bn_apply(x, scale, shift, y2, residual=residual, relu=True)
        assert torch.equal(y, y2)
        pos = (y.reshape(m, c // 32, 32) > 0).to(torch.int64)
        words = (pos << torch.arange(32, dtype=torch.int64, device="cuda")).sum(-1)
        want = torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32)
        assert torch.equal(bits, want)
    # the residual as the input of a second BatchNorm (res_scale / res_shift): what two passes make, bit for bit in fp32
    # (in bf16 the intermediate is not rounded to bf16: compared to that rounding)
    rs = torch.rand(c, device="cuda", generator=g) + 0.5
    rh = torch.randn(c, device="cuda", generator=g) * 0.3
    mid = torch.empty_like(x)
    hip_ops.bn_apply(res, rs, rh, mid)
    want = torch.empty_like(x)
    hip_ops.bn_apply(x, scale, shift, want, residual=mid, relu=True)
    got = torch.empty_like(x)
    hip_ops.bn_apply(x, scale, shift, got, residual=res, relu=True, res_scale=rs, res_shift=rh)
    if dtype == torch.float32:
        assert torch.equal(got, want)
    else:
        torch.testing.assert_close(got.float(), want.float(), rtol=1e-2, atol=3e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,pad", [((3, 23, 18, 64), 1), ((2, 16, 16, 64), 0), ((2, 9, 11, 128), 1), ((1, 12, 12, 24), 1)],
                         ids=lambda v: str(v).replace(" ", ""))
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_maxpool_bwd_with_the_stem_batchnorm_reduction(shape, pad, dtype):
    """maxpool_bwd_bn: the same dx as maxpool_bwd, bit for bit, and slab rows that sum to what bn_bwd_reduce (masked from z)
    makes of that dx - checked against a float64 sum.  C = 24 (6 or 3 chunks of 16 bytes per row, no divisor of 256) is not taken."""
    from distributedpytorch_amd.ops import hip_ops as H, ref_ops as R
    n, h, w, c = shape
    k, s = 3, 2
    ho, wo = R.pool_out_size(h, k, s, pad, False), R.pool_out_size(w, k, s, pad, False)
    g = torch.Generator(device="cuda").manual_seed(h * w + c)
    z = torch.randn(shape, device="cuda", generator=g).to(dtype)
    scale = torch.rand(c, device="cuda", generator=g) + 0.5
    shift = torch.randn(c, device="cuda", generator=g) * 0.3
    mean = torch.randn(c, device="cuda", generator=g) * 0.1
    invstd = torch.rand(c, device="cuda", generator=g) + 0.5
    yp = torch.empty(n, ho, wo, c, device="cuda", dtype=dtype)
    idx = torch.empty(n, ho, wo, c, device="cuda", dtype=torch.uint8)
    H.maxpool_fwd(z, yp, idx, (k, k), (s, s), (pad, pad), scale=scale, shift=shift)
    dyp = torch.randn(n, ho, wo, c, device="cuda", generator=g).to(dtype)
    dx0 = torch.empty_like(z)
    H.maxpool_bwd(dyp, idx, dx0, (k, k), (s, s), (pad, pad))
    dx1 = torch.full_like(z, float("nan"))
    slab = torch.full((2048, 2, c), float("nan"), device="cuda")
    rows = H.maxpool_bwd_bn(dyp, idx, dx1, (k, k), (s, s), (pad, pad), z, scale, shift, mean, invstd, slab)
    if 256 % (c // (4 if dtype == torch.float32 else 8)):      # C = 24: 6 (fp32) or 3 (bf16) chunks per row
        assert rows == 0
        return
    assert 1 <= rows <= 2048
    assert torch.equal(dx0, dx1)
    d = dx0.double() * ((z.float() * scale + shift) > 0)
    want_s = d.sum((0, 1, 2))
    want_q = (d * ((z.double() - mean.double()) * invstd.double())).sum((0, 1, 2))
    got = slab[:rows].double().sum(0)
    assert torch.isnan(slab[rows:]).all()
    mag = d.abs().sum((0, 1, 2)).max().item()
    assert (got[0] - want_s).abs().max().item() <= 1e-5 * mag
    assert (got[1] - want_q).abs().max().item() <= 3e-5 * mag


@pytest.mark.gpu
@pytest.mark.parametrize("m,c", [(1000, 64), (513, 256), (77, 2048)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_bn_bwd_dx_of_two_batchnorms_in_one_pass(m, c, dtype):
    """bn_bwd_dx2 = two bn_bwd_dx launches over the same gradient, bit for bit (dx written over x, as the layers do)."""
    from distributedpytorch_amd.ops import hip_ops
    g = torch.Generator(device="cuda").manual_seed(m + c)
    dz = torch.randn(m, 1, 1, c, device="cuda", generator=g).to(dtype)
    x1 = torch.randn(m, 1, 1, c, device="cuda", generator=g).to(dtype)
    x2 = torch.randn(m, 1, 1, c, device="cuda", generator=g).to(dtype)
    c1 = torch.randn(3, c, device="cuda", generator=g)
    c2 = torch.randn(3, c, device="cuda", generator=g)
    w1, w2 = torch.empty_like(x1), torch.empty_like(x2)
    hip_ops.bn_bwd_dx(dz, x1, c1, w1)
    hip_ops.bn_bwd_dx(dz, x2, c2, w2)
    a, b = x1.clone(), x2.clone()
    hip_ops.bn_bwd_dx2(dz, a, c1, a, b, c2, b)
    assert torch.equal(a, w1) and torch.equal(b, w2)
