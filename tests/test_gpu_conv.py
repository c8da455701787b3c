"""T2 kernel correctness: HIP implicit-GEMM conv fprop / dgrad / wgrad vs a plain PyTorch fp32 reference."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (N, H, W, Cin, Cout, k, stride, pad)
SHAPES = [
    (4, 56, 56, 64, 64, 3, 1, 1),      # resnet layer1
    (4, 56, 56, 64, 128, 3, 2, 1),     # layer2.0.conv1
    (4, 56, 56, 64, 128, 1, 2, 0),     # downsample
    (2, 28, 28, 128, 128, 3, 1, 1),
    (2, 14, 14, 256, 256, 3, 1, 1),
    (8, 7, 7, 512, 512, 3, 1, 1),
    (2, 32, 32, 8, 64, 7, 2, 3),       # stem (Cin padded to 8)
    (2, 56, 56, 64, 256, 1, 1, 0),     # bottleneck expand
    (4, 28, 28, 128, 32, 3, 1, 1),     # DenseNet growth conv: 32 filters (wgrad 32 x 192 tile)
    (2, 17, 17, 64, 24, 3, 1, 1),      # 24 filters, Ng = 576 (wgrad 32 x 192 tile with padded rows)
    (2, 9, 9, 32, 32, 5, 1, 2),        # 32 filters, Ng = 800 (wgrad 32 x 256 tile)
    (3, 17, 13, 24, 40, 5, 1, 2),      # odd sizes, channels not multiples of the tile
    (2, 20, 20, 16, 48, 11, 4, 2),     # alexnet-like stem
    (64, 1, 1, 512, 16, 1, 1, 0),      # linear as 1x1 conv
    (65, 32, 32, 128, 128, 3, 1, 1),   # 520 big tiles = 1.02 rounds: tail-split launch (128x128 body + 64x64 tail)
    (260, 32, 32, 128, 128, 3, 2, 1),  # tail split in fprop and in each stride-2 dgrad parity class
]


def _ref(x, w, stride, pad):
    # x NHWC, w KRSC -> NCHW reference in fp32
    return F.conv2d(x.permute(0, 3, 1, 2).float(), w.permute(0, 3, 1, 2).float(), stride=stride, padding=pad)


def _tol(dtype, kred):
    if dtype == torch.float32:
        return 2e-5 * max(1.0, kred ** 0.5), 1e-5
    return 2e-2, 2e-2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", SHAPES)
def test_conv_fprop_dgrad_wgrad(shape, dtype, fp32_mode):
    from distributedpytorch_amd.ops import hip_conv
    if dtype == torch.bfloat16 and fp32_mode == "exact":
        pytest.skip("the fp32 arithmetic mode does not affect the bf16 kernels")
    N, H, W, C, K, k, s, pd = shape
    g = torch.Generator(device="cuda").manual_seed(1234)
    x = torch.randn(N, H, W, C, device="cuda", generator=g).to(dtype)
    w = (torch.randn(K, k, k, C, device="cuda", generator=g) * (k * k * C) ** -0.5).to(dtype)
    Ho, Wo = hip_conv.out_size(H, k, s, pd), hip_conv.out_size(W, k, s, pd)
    y = torch.full((N, Ho, Wo, K), float("nan"), device="cuda", dtype=dtype)
    rows = hip_conv.stats_rows(N * Ho * Wo, K, dtype)
    stats = torch.zeros(rows, 2, K, device="cuda")
    hip_conv.conv_fprop(x, w, y, (s, s), (pd, pd), stats=stats)
    xr = x.float().requires_grad_(True)
    wr = w.float().requires_grad_(True)
    yr = _ref(xr, wr, s, pd)
    rtol, atol = _tol(dtype, k * k * C)
    torch.testing.assert_close(y.float().permute(0, 3, 1, 2), yr.detach(), rtol=rtol, atol=atol * 10)
    # fused BN statistics
    st = stats.sum(0)
    yf = y.float().reshape(-1, K)
    torch.testing.assert_close(st[0], yf.sum(0), rtol=1e-3, atol=1e-2)
    torch.testing.assert_close(st[1], (yf * yf).sum(0), rtol=1e-3, atol=1e-2)

    dy = torch.randn(N, Ho, Wo, K, device="cuda", generator=g).to(dtype)
    yr.backward(dy.float().permute(0, 3, 1, 2))
    # dgrad
    wt = w.permute(3, 1, 2, 0).contiguous()
    dx = torch.full((N, H, W, C), float("nan"), device="cuda", dtype=dtype)
    hip_conv.conv_dgrad(dy, wt, dx, (s, s), (pd, pd))
    torch.testing.assert_close(dx.float(), xr.grad, rtol=rtol, atol=atol * 10)
    # wgrad (split-K with workspace, then single pass accumulate)
    dw = torch.full((K, k, k, C), float("nan"), device="cuda")
    ws = torch.empty(hip_conv.wgrad_workspace_numel(K, k * k * C), device="cuda")
    hip_conv.conv_wgrad(dy, x, dw, (k, k), (s, s), (pd, pd), workspace=ws)
    ref_dw = wr.grad
    scale = ref_dw.abs().max().item()
    torch.testing.assert_close(dw, ref_dw, rtol=rtol, atol=max(atol * 10, scale * (3e-2 if dtype == torch.bfloat16 else 1e-5)))
    dw2 = torch.ones_like(dw)
    hip_conv.conv_wgrad(dy, x, dw2, (k, k), (s, s), (pd, pd), workspace=None, accumulate=True)
    torch.testing.assert_close(dw2 - 1.0, ref_dw, rtol=rtol, atol=max(atol * 10, scale * (3e-2 if dtype == torch.bfloat16 else 2e-5)))


X6Q_SHAPES = [
    (2, 14, 14, 256, 256, 3, 1, 1),    # 256 x 128 tiles
    (2, 14, 14, 120, 480, 3, 1, 1),    # 256 x 128 tiles, ragged in both dimensions (K = 480, Ng = 1080)
    (3, 28, 28, 136, 120, 3, 1, 1),    # 128 x 256 tiles, ragged in both dimensions (K = 120, Ng = 1224)
    (2, 28, 28, 512, 128, 1, 1, 0),    # 128 x 256 tiles, 1x1
    (2, 28, 28, 256, 512, 1, 2, 0),    # 1x1 stride 2
    (1, 9, 9, 64, 256, 3, 2, 1),       # stride 2 with padding; 25 output pixels: less than ONE 32-pixel K-tile
    (5, 7, 7, 512, 2048, 1, 1, 0),     # 245 pixels: 7.7 K-tiles, odd count of K-tiles per split
]


@pytest.mark.parametrize("shape", X6Q_SHAPES)
def test_conv_wgrad_x6q_kernel(shape):
    """conv_wgrad_x6q_kernel (fp32 x6, 8 waves, plane tiles double-buffered in LDS, staging interleaved with the MFMAs)
    against the PyTorch reference and against the 4-wave kernel; split-K with a workspace, and one pass accumulating."""
    from distributedpytorch_amd.ops import hip_conv, native
    from distributedpytorch_amd.ops.native import set_fp32_mode
    lib = native.hip_lib()
    N, H, W, C, K, k, s, pd = shape
    g = torch.Generator(device="cuda").manual_seed(2468)
    x = torch.randn(N, H, W, C, device="cuda", generator=g)
    Ho, Wo = hip_conv.out_size(H, k, s, pd), hip_conv.out_size(W, k, s, pd)
    dy = torch.randn(N, Ho, Wo, K, device="cuda", generator=g)
    wr = torch.zeros(K, C, k, k, device="cuda", dtype=torch.float64, requires_grad=True)
    F.conv2d(x.permute(0, 3, 1, 2).double(), wr, stride=s, padding=pd).backward(dy.permute(0, 3, 1, 2).double())
    ref = wr.grad.permute(0, 2, 3, 1).float()
    scale = ref.abs().max().item()
    prev_mode = set_fp32_mode("x6")
    prev = lib.dpa_wgrad_knob(1, 1)
    try:
        ws = torch.empty(hip_conv.wgrad_workspace_numel(K, k * k * C), device="cuda")
        dw = torch.full((K, k, k, C), float("nan"), device="cuda")
        hip_conv.conv_wgrad(dy, x, dw, (k, k), (s, s), (pd, pd), workspace=ws)
        torch.testing.assert_close(dw, ref, rtol=1e-5, atol=scale * 2e-6)
        dw1 = torch.ones_like(dw)
        hip_conv.conv_wgrad(dy, x, dw1, (k, k), (s, s), (pd, pd), workspace=None, accumulate=True)
        torch.testing.assert_close(dw1 - 1.0, ref, rtol=1e-5, atol=scale * 4e-6)
        lib.dpa_wgrad_knob(1, 0)
        dw0 = torch.full((K, k, k, C), float("nan"), device="cuda")
        hip_conv.conv_wgrad(dy, x, dw0, (k, k), (s, s), (pd, pd), workspace=ws)
        torch.testing.assert_close(dw, dw0, rtol=1e-5, atol=scale * 2e-6)
    finally:
        lib.dpa_wgrad_knob(1, prev)
        set_fp32_mode(prev_mode)


def test_conv_channel_slice_and_epilogue():
    """zero-copy concat: read a channel slice, write into a channel slice, fused bias+relu+addend."""
    from distributedpytorch_amd.ops import hip_conv
    dtype = torch.float32
    g = torch.Generator(device="cuda").manual_seed(7)
    buf_in = torch.randn(2, 12, 12, 48, device="cuda", generator=g)
    x = buf_in[..., 16:40]
    w = torch.randn(32, 3, 3, 24, device="cuda", generator=g) * 0.1
    b = torch.randn(32, device="cuda", generator=g)
    buf_out = torch.zeros(2, 12, 12, 64, device="cuda")
    y = buf_out[..., 32:64]
    add = torch.randn(2, 12, 12, 64, device="cuda", generator=g)[..., 32:64]
    hip_conv.conv_fprop(x, w, y, (1, 1), (1, 1), bias=b, relu=True, addend=add)
    ref = F.conv2d(x.permute(0, 3, 1, 2), w.permute(0, 3, 1, 2), bias=b, padding=1).permute(0, 2, 3, 1) + add
    torch.testing.assert_close(y, ref.clamp_min(0), rtol=1e-4, atol=1e-4)
    assert buf_out[..., :32].abs().max().item() == 0.0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_splitk_skinny_gemm(dtype, fp32_mode):
    """Linear-shaped problems (few output tiles, long reduction) take the split-K path: partial sums in a workspace,
    bias + addend + ReLU in the reduce kernel.  Compared with the same op forced through the single-pass kernel's
    reference (plain fp32 matmul)."""
    from distributedpytorch_amd.ops import hip_conv
    g = torch.Generator(device="cuda").manual_seed(11)
    N, K, C = 64, 4608, 520                      # 520 outputs: not a multiple of the 64-wide tile
    x = torch.randn(N, 1, 1, K, device="cuda", generator=g).to(dtype)
    w = (torch.randn(C, 1, 1, K, device="cuda", generator=g) * K ** -0.5).to(dtype)
    b = torch.randn(C, device="cuda", generator=g)
    add = torch.randn(N, 1, 1, C, device="cuda", generator=g).to(dtype)
    y = torch.full((N, 1, 1, C), float("nan"), device="cuda", dtype=dtype)
    hip_conv.conv_fprop(x, w, y, (1, 1), (0, 0), bias=b, relu=True, addend=add)
    ref = (x.float().view(N, K) @ w.float().view(C, K).t() + b + add.float().view(N, C)).clamp_min(0)
    tol = dict(rtol=2e-4, atol=2e-4) if dtype == torch.float32 else dict(rtol=2e-2, atol=3e-2)
    torch.testing.assert_close(y.float().view(N, C), ref, **tol)
    # dgrad of the same layer: dX[N, K] = dY[N, C] @ W[C, K]  (K outputs, reduction over C)
    dy = torch.randn(N, 1, 1, C, device="cuda", generator=g).to(dtype)
    wt = w.permute(3, 1, 2, 0).contiguous()
    dx = torch.full((N, 1, 1, K), float("nan"), device="cuda", dtype=dtype)
    hip_conv.conv_dgrad(dy, wt, dx, (1, 1), (0, 0))
    torch.testing.assert_close(dx.float().view(N, K), dy.float().view(N, C) @ w.float().view(C, K), **tol)


# (N, H, W, Cin, Cout, k, stride, pad): shapes routed to the pre-split x6 kernel (Cin % 32 == 0), with ragged tiles
PRESPLIT_SHAPES = [
    (4, 56, 56, 64, 64, 3, 1, 1),      # BN = 64 variant
    (3, 28, 28, 128, 128, 3, 1, 1),    # BN = 128 variant, M = 2352 (18.4 tiles: ragged last tile)
    (2, 28, 28, 64, 128, 3, 2, 1),     # stride 2: dgrad parity classes
    (2, 14, 14, 256, 1024, 1, 1, 0),   # 1x1 expand
    (5, 14, 14, 1024, 256, 1, 1, 0),   # 1x1 reduce, long reduction
    (2, 17, 13, 96, 72, 3, 1, 1),      # Cout neither a multiple of 64 nor of 128, odd spatial sizes
    (2, 28, 28, 256, 512, 1, 2, 0),    # downsample: 1x1 stride 2 (three of four dgrad classes have no taps)
    (130, 1, 1, 512, 160, 1, 1, 0),    # linear-shaped
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(3, 28, 28, 128, 128, 3, 1, 1), (2, 14, 14, 256, 1024, 1, 1, 0), (130, 1, 1, 512, 160, 1, 1, 0),
                                   (2, 17, 13, 24, 40, 3, 2, 1)])
def test_conv_fprop_output_scale_epilogue(shape, dtype, fp32_mode):
    """ConvParams::oscale: out = relu(conv * scale[c] + bias[c] + addend) - the inference BatchNorm folded into the epilogue -
    in the pre-split kernel, the general kernel and its split-K reduction (the linear-shaped problem)."""
    from distributedpytorch_amd.ops import hip_conv
    if dtype == torch.bfloat16 and fp32_mode == "exact":
        pytest.skip("the fp32 arithmetic mode does not affect the bf16 kernels")
    N, H, W, C, K, k, s, pd = shape
    g = torch.Generator(device="cuda").manual_seed(99)
    x = torch.randn(N, H, W, C, device="cuda", generator=g).to(dtype)
    w = (torch.randn(K, k, k, C, device="cuda", generator=g) * (k * k * C) ** -0.5).to(dtype)
    sc = torch.rand(K, device="cuda", generator=g) + 0.5
    b = torch.randn(K, device="cuda", generator=g)
    Ho, Wo = hip_conv.out_size(H, k, s, pd), hip_conv.out_size(W, k, s, pd)
    add = torch.randn(N, Ho, Wo, K, device="cuda", generator=g).to(dtype)
    w3 = None
    if dtype == torch.float32 and C % 32 == 0:
        w3 = hip_conv.split3_planes(w, torch.empty(3 * w.numel(), device="cuda", dtype=torch.bfloat16))
    y = torch.full((N, Ho, Wo, K), float("nan"), device="cuda", dtype=dtype)
    hip_conv.conv_fprop(x, w, y, (s, s), (pd, pd), bias=b, relu=True, addend=add, w3=w3, oscale=sc)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), None, stride=s, padding=pd).permute(0, 2, 3, 1)
    ref = (ref * sc + b + add.float()).clamp_min(0)
    rtol, atol = _tol(dtype, k * k * C)
    torch.testing.assert_close(y.float(), ref, rtol=rtol, atol=atol * 10)


@pytest.fixture(params=["short", "narrow", "wide", "wide-plain"])
def x6p_variant(request):
    """Steer the (small) test problems to each tile shape of the pre-split kernel: 64 x 64 (three blocks per CU), 128 x 64
    and 128 x 128 (two blocks per CU); production picks by problem size (DPA_X6P_NARROW_BELOW, DPA_X6P_SHORT_BELOW).
    The 128 x 128 tile with both K loops: the software-pipelined one for EVERY reduction length (production: four K-tiles
    and more, DPA_X6P_PIPE_MIN_KT) and, "-plain", the double-buffered one the other tile shapes always use."""
    from distributedpytorch_amd.ops import native
    lib = native.hip_lib()
    big = 1 << 30
    tile, _, loop = request.param.partition("-")
    narrow_below, short_below = {"short": (big, big), "narrow": (big, 0), "wide": (0, 0)}[tile]
    prev = lib.dpa_x6p_knob(1, narrow_below), lib.dpa_x6p_knob(2, short_below), lib.dpa_x6p_knob(3, 0 if loop == "plain" else 1)
    yield tile
    lib.dpa_x6p_knob(1, prev[0]); lib.dpa_x6p_knob(2, prev[1]); lib.dpa_x6p_knob(3, prev[2])


@pytest.mark.parametrize("shape", PRESPLIT_SHAPES)
def test_conv_presplit_x6_kernel(shape, x6p_variant):
    """csrc/hip/conv_x6p.hip (fp32 through the bf16 matrix pipe with the filter pre-split into three bf16 planes, waves
    stacked along M) against the fp32 PyTorch reference: fprop with the whole fused epilogue (bias, shifted BN statistics,
    addend, ReLU, channel-sliced destination) and dgrad (fresh and accumulating in place)."""
    from distributedpytorch_amd.ops import hip_conv
    from distributedpytorch_amd.ops.native import set_fp32_mode
    N, H, W, C, K, k, s, pd = shape
    g = torch.Generator(device="cuda").manual_seed(4321)
    x = torch.randn(N, H, W, C, device="cuda", generator=g)
    w = torch.randn(K, k, k, C, device="cuda", generator=g) * (k * k * C) ** -0.5
    b = torch.randn(K, device="cuda", generator=g)
    Ho, Wo = hip_conv.out_size(H, k, s, pd), hip_conv.out_size(W, k, s, pd)
    M = N * Ho * Wo
    prev = set_fp32_mode("x6")
    try:
        w3 = hip_conv.split3_planes(w, torch.empty(3 * w.numel(), device="cuda", dtype=torch.bfloat16))
        # plain
        y = torch.full((N, Ho, Wo, K), float("nan"), device="cuda")
        stats = torch.full(((M + 63) // 64, 2, K), float("nan"), device="cuda")
        shift = torch.randn(K, device="cuda", generator=g) * 0.1
        hip_conv.conv_fprop(x, w, y, (s, s), (pd, pd), bias=b, stats=stats, stats_shift=shift, w3=w3)
        rows = hip_conv.last_stats_rows[0]
        tile_rows = 64 if x6p_variant == "short" else 128
        assert rows == (M + tile_rows - 1) // tile_rows     # one slab row per M-tile of that variant: the pre-split kernel ran
        yr = F.conv2d(x.permute(0, 3, 1, 2), w.permute(0, 3, 1, 2), b, stride=s, padding=pd).permute(0, 2, 3, 1)
        rtol, atol = _tol(torch.float32, k * k * C)
        torch.testing.assert_close(y, yr, rtol=rtol, atol=atol * 10)
        st = stats[:rows].sum(0)
        yc = (y - shift).reshape(-1, K)
        torch.testing.assert_close(st[0], yc.sum(0), rtol=1e-3, atol=1e-2)
        torch.testing.assert_close(st[1], (yc * yc).sum(0), rtol=1e-3, atol=1e-2)
        # fused addend + ReLU into a channel slice of a wider buffer
        buf = torch.zeros(N, Ho, Wo, K + 16, device="cuda")
        add = torch.randn(N, Ho, Wo, K + 16, device="cuda", generator=g)
        hip_conv.conv_fprop(x, w, buf[..., 8:8 + K], (s, s), (pd, pd), bias=b, relu=True, addend=add[..., 8:8 + K], w3=w3)
        torch.testing.assert_close(buf[..., 8:8 + K], (yr + add[..., 8:8 + K]).clamp_min(0), rtol=rtol, atol=atol * 10)
        assert buf[..., :8].abs().max() == 0 and buf[..., 8 + K:].abs().max() == 0
        # dgrad: fresh, then accumulating in place
        if K % 32 == 0:
            dy = torch.randn(N, Ho, Wo, K, device="cuda", generator=g)
            wt = w.permute(3, 1, 2, 0).contiguous()
            wt3 = hip_conv.split3_planes(wt, torch.empty(3 * wt.numel(), device="cuda", dtype=torch.bfloat16))
            ref_dx = torch.nn.grad.conv2d_input((N, C, H, W), w.permute(0, 3, 1, 2), dy.permute(0, 3, 1, 2), stride=s,
                                                padding=pd).permute(0, 2, 3, 1)
            dx = torch.full((N, H, W, C), float("nan"), device="cuda")
            hip_conv.conv_dgrad(dy, wt, dx, (s, s), (pd, pd), wt3=wt3)
            rt, at = _tol(torch.float32, k * k * K)
            torch.testing.assert_close(dx, ref_dx, rtol=rt, atol=at * 10)
            acc = torch.randn(N, H, W, C, device="cuda", generator=g)
            acc0 = acc.clone()
            hip_conv.conv_dgrad(dy, wt, acc, (s, s), (pd, pd), addend=acc, wt3=wt3)
            torch.testing.assert_close(acc, acc0 + ref_dx, rtol=rt, atol=at * 10)
    finally:
        set_fp32_mode(prev)


# (N, H, W, Cin(of the forward conv = channels of dx), Cout(channels of dy), k, stride, pad)
BNR_SHAPES = [
    (4, 28, 28, 128, 128, 3, 1, 1),     # pre-split x6 kernel (fp32), 128-wide tile
    (4, 28, 28, 64, 256, 1, 1, 0),      # dC = 64: the narrow x6p / 256 x 64 tiles
    (3, 30, 30, 128, 128, 3, 2, 1),     # stride 2: four parity classes, each with its own slab rows
    (2, 17, 13, 24, 40, 3, 1, 1),       # general kernel (Cout of dy not a multiple of 32), partial tiles
    (65, 32, 32, 128, 128, 3, 1, 1),    # tail-split launch
]


def _pack_mask_bits(y):
    """[..., C] -> int32 [rows, C / 32], bit b of word w = (y[row, 32 w + b] > 0) (the layout bn_apply(mask_bits=) writes)."""
    c = y.shape[-1]
    pos = (y.reshape(-1, c // 32, 32) > 0).to(torch.int64)
    words = (pos << torch.arange(32, dtype=torch.int64, device=y.device)).sum(-1)
    return torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32).contiguous()


@pytest.mark.parametrize("mask", ["y", "ybits", "x", "none"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", BNR_SHAPES)
def test_conv_dgrad_fused_bn_backward_reduce(shape, dtype, mask, fp32_mode, x6p_variant):
    """conv_dgrad(bn_reduce=...): the epilogue masks the gradient it stores and writes the partial sums of the
    BatchNorm-backward reduction; compared with the unfused dgrad followed by the plain formulas in fp64."""
    from distributedpytorch_amd.ops import hip_conv
    if dtype == torch.bfloat16 and fp32_mode == "exact":
        pytest.skip("the fp32 arithmetic mode does not affect the bf16 kernels")
    if x6p_variant != "narrow" and not (dtype == torch.float32 and fp32_mode == "x6"):
        pytest.skip("the tile variants exist in the pre-split fp32 kernel only")
    N, H, W, C, K, k, s, pd = shape
    if mask == "ybits" and C % 32:
        pytest.skip("bit masks need a multiple of 32 channels")
    g = torch.Generator(device="cuda").manual_seed(99)
    Ho, Wo = hip_conv.out_size(H, k, s, pd), hip_conv.out_size(W, k, s, pd)
    dy = torch.randn(N, Ho, Wo, K, device="cuda", generator=g).to(dtype)
    wt = (torch.randn(C, k, k, K, device="cuda", generator=g) * (k * k * K) ** -0.5).to(dtype)
    z = (torch.randn(N, H, W, C, device="cuda", generator=g) * 2 + 0.5).to(dtype)      # the BatchNorm's input
    mean = torch.randn(C, device="cuda", generator=g) * 0.3 + 0.5
    invstd = torch.rand(C, device="cuda", generator=g) + 0.5
    msc = torch.randn(C, device="cuda", generator=g)
    msh = torch.randn(C, device="cuda", generator=g) * 0.2
    y = torch.randn(N, H, W, C, device="cuda", generator=g).to(dtype)                  # its (residual + ReLU) output
    addend = torch.randn(N, H, W, C, device="cuda", generator=g).to(dtype)
    wt3 = None
    if dtype == torch.float32 and fp32_mode == "x6" and K % 32 == 0:
        wt3 = hip_conv.split3_planes(wt, torch.empty(3 * wt.numel(), device="cuda", dtype=torch.bfloat16))
    for use_addend in (False, True):
        plain = torch.empty(N, H, W, C, device="cuda", dtype=dtype)
        if use_addend:
            plain.copy_(addend)
        hip_conv.conv_dgrad(dy, wt, plain, (s, s), (pd, pd), addend=plain if use_addend else None, wt3=wt3)
        if mask in ("y", "ybits"):
            keep = y > 0
        elif mask == "x":
            keep = (z.float() * msc + msh) > 0
        else:
            keep = torch.ones_like(z, dtype=torch.bool)
        want = torch.where(keep, plain, torch.zeros_like(plain))
        rows = hip_conv.bn_reduce_rows(N * H * W, (s, s))
        slab = torch.full((rows, 2, C), float("nan"), device="cuda")
        fused = torch.full((N, H, W, C), float("nan"), device="cuda", dtype=dtype)
        if use_addend:
            fused.copy_(addend)
        # (with the bit mask: also a SECOND BatchNorm under the same ReLU - sum dz * xhat2 into a slab of its own - where the
        # launch takes the pre-split fp32 kernel; any other launch must refuse it before anything runs)
        second = None
        if mask == "ybits":
            z2 = (torch.randn(N, H, W, C, device="cuda", generator=g) * 1.5 - 0.3).to(dtype)
            mean2 = torch.randn(C, device="cuda", generator=g) * 0.3 - 0.3
            invstd2 = torch.rand(C, device="cuda", generator=g) + 0.4
            second = (z2, mean2, invstd2, torch.full((rows, 2, C), float("nan"), device="cuda"))
        # (with bits, the y handed over is the NEGATED one: a kernel that read it instead of the bits would mask the wrong half)
        reduce_args = (z, (-y if mask == "ybits" else y) if mask in ("y", "ybits") else None, mean, invstd,
                       msc if mask == "x" else None, msh if mask == "x" else None, slab) \
            + ((_pack_mask_bits(y),) if mask == "ybits" else ())
        did_second = False
        if second is not None:
            try:
                used = hip_conv.conv_dgrad(dy, wt, fused, (s, s), (pd, pd), addend=fused if use_addend else None, wt3=wt3,
                                           bn_reduce=reduce_args + (second,))
                did_second = True
            except hip_conv.SecondBnUnsupported:
                assert not (dtype == torch.float32 and fp32_mode == "x6" and wt3 is not None and C > 32), "the pre-split kernel refused"
                torch.cuda.synchronize()
                assert torch.isnan(slab).all()                     # nothing was launched
        if not did_second:
            used = hip_conv.conv_dgrad(dy, wt, fused, (s, s), (pd, pd), addend=fused if use_addend else None, wt3=wt3,
                                       bn_reduce=reduce_args)
        assert 0 < used <= rows
        assert torch.equal(fused, want)                      # same arithmetic, masked
        got = slab[:used].double().sum(0)
        assert torch.isfinite(got).all()
        wd = want.double().reshape(-1, C)
        xh = (z.double().reshape(-1, C) - mean.double()) * invstd.double()
        ref_s, ref_q = wd.sum(0), (wd * xh).sum(0)
        scale_s = wd.abs().sum(0).clamp_min(1e-6)
        scale_q = (wd * xh).abs().sum(0).clamp_min(1e-6)
        assert ((got[0] - ref_s).abs() / scale_s).max().item() < 2e-6
        assert ((got[1] - ref_q).abs() / scale_q).max().item() < 2e-6
        if did_second:
            got2 = second[3][:used].double().sum(0)
            xh2 = (z2.double().reshape(-1, C) - mean2.double()) * invstd2.double()
            ref_q2 = (wd * xh2).sum(0)
            scale_q2 = (wd * xh2).abs().sum(0).clamp_min(1e-6)
            assert torch.equal(second[3][:used, 0], slab[:used, 0])          # the same sum dz, row by row
            assert ((got2[1] - ref_q2).abs() / scale_q2).max().item() < 2e-6


@pytest.fixture
def x6s_small():
    """Let small problems take the persistent streaming kernel (conv_x6s.hip): 16 resident blocks instead of 512."""
    from distributedpytorch_amd.ops import native
    lib = native.hip_lib()
    prev = lib.dpa_x6s_knob(2, 16), lib.dpa_x6p_min_tiles(1), lib.dpa_x6p_knob(1, 0), lib.dpa_x6p_knob(2, 0)
    lib.x6s_slots_expected = lambda tiles_n: max(8, (16 // tiles_n) & ~7)
    yield lib
    lib.dpa_x6s_knob(2, prev[0]); lib.dpa_x6p_min_tiles(prev[1]); lib.dpa_x6p_knob(1, prev[2]); lib.dpa_x6p_knob(2, prev[3])


# (N, H, W, Cout): 1x1 / stride 1 on 64 input channels; M = N H W is not a multiple of the 128-row tile
X6S_SHAPES = [(6, 37, 29, 64), (6, 37, 29, 256), (9, 32, 32, 128), (4, 30, 30, 512)]


@pytest.mark.parametrize("shape", X6S_SHAPES)
def test_conv_streaming_x6_kernel(shape, x6s_small):
    """csrc/hip/conv_x6s.hip (persistent blocks, filter planes resident in LDS, next activation tile in flight, stores straight
    from the accumulators) against the fp32 PyTorch reference and against conv_x6p.hip's kernel, whose bits it must
    reproduce: fprop with bias + shifted BatchNorm statistics, addend + ReLU into a channel slice, dgrad fresh and
    accumulating in place."""
    from distributedpytorch_amd.ops import hip_conv
    from distributedpytorch_amd.ops.native import set_fp32_mode
    lib = x6s_small
    N, H, W, K = shape
    C = 64
    g = torch.Generator(device="cuda").manual_seed(77)
    x = torch.randn(N, H, W, C, device="cuda", generator=g)
    w = torch.randn(K, 1, 1, C, device="cuda", generator=g) * C ** -0.5
    b = torch.randn(K, device="cuda", generator=g)
    shift = torch.randn(K, device="cuda", generator=g) * 0.1
    M = N * H * W
    prev = set_fp32_mode("x6")
    try:
        w3 = hip_conv.split3_planes(w, torch.empty(3 * w.numel(), device="cuda", dtype=torch.bfloat16))
        add = torch.randn(N, H, W, K + 16, device="cuda", generator=g)
        out = {}
        for on in (1, 0):
            lib.dpa_x6s_knob(1, on)
            y = torch.full((N, H, W, K), float("nan"), device="cuda")
            stats = torch.full(((M + 63) // 64, 2, K), float("nan"), device="cuda")
            hip_conv.conv_fprop(x, w, y, (1, 1), (0, 0), bias=b, stats=stats, stats_shift=shift, w3=w3)
            rows = hip_conv.last_stats_rows[0]
            buf = torch.zeros(N, H, W, K + 16, device="cuda")
            hip_conv.conv_fprop(x, w, buf[..., 8:8 + K], (1, 1), (0, 0), bias=b, relu=True, addend=add[..., 8:8 + K], w3=w3)
            out[on] = (y, stats[:rows].double().sum(0), rows, buf)
        tilesN = 1 if K == 64 else K // 128
        assert out[1][2] == lib.x6s_slots_expected(tilesN) and out[0][2] == (M + 127) // 128      # the streaming kernel ran / did not
        y, st, _, buf = out[1]
        assert torch.equal(y, out[0][0]) and torch.equal(buf, out[0][3])
        yr = F.conv2d(x.permute(0, 3, 1, 2), w.permute(0, 3, 1, 2), b).permute(0, 2, 3, 1)
        rtol, atol = _tol(torch.float32, C)
        torch.testing.assert_close(y, yr, rtol=rtol, atol=atol * 10)
        yc = (y.double() - shift.double()).reshape(-1, K)
        torch.testing.assert_close(st[0], yc.sum(0), rtol=1e-5, atol=1e-3)
        torch.testing.assert_close(st[1], (yc * yc).sum(0), rtol=1e-5, atol=1e-3)
        torch.testing.assert_close(buf[..., 8:8 + K], (yr + add[..., 8:8 + K]).clamp_min(0), rtol=rtol, atol=atol * 10)
        assert buf[..., :8].abs().max() == 0 and buf[..., 8 + K:].abs().max() == 0
        # dgrad of a 1x1 conv K -> 64 ... is the same GEMM with the transposed filter: dy has 64 channels here
        dy = torch.randn(N, H, W, C, device="cuda", generator=g)
        wf = torch.randn(C, 1, 1, K, device="cuda", generator=g) * K ** -0.5         # forward filter of a K -> 64 conv
        wt = wf.permute(3, 1, 2, 0).contiguous()                                     # [K][1][1][64]
        wt3 = hip_conv.split3_planes(wt, torch.empty(3 * wt.numel(), device="cuda", dtype=torch.bfloat16))
        ref_dx = torch.nn.grad.conv2d_input((N, K, H, W), wf.permute(0, 3, 1, 2), dy.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
        res = {}
        for on in (1, 0):
            lib.dpa_x6s_knob(1, on)
            dx = torch.full((N, H, W, K), float("nan"), device="cuda")
            hip_conv.conv_dgrad(dy, wt, dx, (1, 1), (0, 0), wt3=wt3)
            acc = add[..., :K].contiguous()
            hip_conv.conv_dgrad(dy, wt, acc, (1, 1), (0, 0), addend=acc, wt3=wt3)
            res[on] = (dx, acc)
        assert torch.equal(res[1][0], res[0][0]) and torch.equal(res[1][1], res[0][1])
        torch.testing.assert_close(res[1][0], ref_dx, rtol=rtol, atol=atol * 10)
        torch.testing.assert_close(res[1][1], add[..., :K] + ref_dx, rtol=rtol, atol=atol * 10)
    finally:
        lib.dpa_x6s_knob(1, 1)
        set_fp32_mode(prev)


def test_resnet50_step_through_the_streaming_kernel(x6s_small):
    """Integration: a whole ResNet-50 training step with the 1x1 layers on 64 channels routed through conv_x6s.hip
    against the same step (fresh model, same seed, same batch) with them on conv_x6p.hip.  The convolution outputs carry
    the same bits (test_conv_streaming_x6_kernel); the BatchNorm statistics are summed in another order, per block
    instead of per M-tile.  What that last-bit difference does to this 53-BatchNorm network on 16 images was measured
    with scripts/diag/x6s_amplification.py: ANY change of the summation order (64-row tiles instead of 128-row tiles of
    conv_x6p.hip itself; the second step of one model, whose statistics shift differs) moves the gradient by the same
    amount in the same direction - 1 - cos = 0.4-2.6e-4, conv1.weight by 0.9-2.3 % of its norm - while a repeated run is
    bit-identical (and torch's own fp32, GPU or CPU, is 1.4 % from an fp64 oracle on that same parameter, exactly like this
    engine: scripts/diag/resnet50_grad_vs_fp64.py - a property of this network on 16 small images).  The bounds below are that level with a margin; this test catches gross errors (a block of strips
    missing, slab rows of the wrong column tile), the bit-level checks are in the kernel test."""
    from distributedpytorch_amd.models.factory import get_model
    from distributedpytorch_amd.ops.native import set_fp32_mode
    from tests.helpers import framework_step, grads_by_name
    lib = x6s_small
    prev = set_fp32_mode("x6")
    try:
        out = {}
        for on in (1, 0):
            torch.manual_seed(5)
            net, _ = get_model("resnet50", 10, device="cuda", dtype=torch.float32)
            x = torch.randn(16, 3, 64, 64)        # layer1 runs on 16x16 maps: 4096 rows = 128 strips, two per resident wave
            y = torch.randint(0, 10, (16,))
            lib.dpa_x6s_knob(1, on)
            logits, loss, _ = framework_step(net, x, y)
            out[on] = (logits, loss, grads_by_name(net))
        assert abs(out[1][1] - out[0][1]) < 1e-4 * max(1.0, abs(out[0][1])), (out[1][1], out[0][1])
        torch.testing.assert_close(out[1][0], out[0][0], rtol=1e-2, atol=1e-3)
        ga = torch.cat([out[1][2][k].double().flatten() for k in out[0][2]])
        gb = torch.cat([g.double().flatten() for g in out[0][2].values()])
        cos = torch.dot(ga, gb).item() / (ga.norm().item() * gb.norm().item())
        gmax = max(g.norm().item() for g in out[0][2].values())
        worst = max((out[1][2][k] - g).norm().item() / gmax for k, g in out[0][2].items())
        assert 1 - cos < 2e-3 and worst < 0.1, (1 - cos, worst)
        # the downsample blocks' compact gradients (nn/layers.py: Conv2d._dgrad) against full-resolution ones, and the ReLU
        # masks as bits (ConvBNAct) against the fused epilogues reading y itself: same bits in every gradient
        from distributedpytorch_amd.nn import layers
        keep = layers._COMPACT_DOWN, layers._RELU_BITS, layers._FUSE_BN2, layers._DEFER_DOWN_BN, layers._BN_DX2
        try:
            res = {}
            for compact, relu_bits, bn2 in ((True, True, False), (False, True, False), (True, False, False), (True, True, True),
                                            (True, True, "own_apply"), (True, True, "two_dx")):
                # "two_dx": second reduction fused, but bn3 and the downsample BatchNorm compute dx in separate passes
                # (ops.bn_bwd_dx2 off): the same fp32 operations per element, so the same bits as (True, True, True)
                layers._BN_DX2 = bn2 != "two_dx"
                # "own_apply": the downsample branches apply their BatchNorm themselves instead of leaving it to the block's
                # last bn_apply (ConvBNAct(defer_apply=True)): the same fp32 operations, so the same bits
                layers._DEFER_DOWN_BN = bn2 != "own_apply"
                layers._COMPACT_DOWN, layers._RELU_BITS, layers._FUSE_BN2 = compact, relu_bits, bn2 in (True, "two_dx")
                torch.manual_seed(5)
                net, _ = get_model("resnet50", 10, device="cuda", dtype=torch.float32)
                made, with_bits, with_bn2, dx2 = [], [], [], []
                orig, orig_dx2 = layers.ops.conv_dgrad, layers.ops.bn_bwd_dx2

                def spy_dx2(*a, **kw):
                    dx2.append(1)
                    return orig_dx2(*a, **kw)

                def spy(*a, **kw):
                    made.append(kw.get("addend2") is not None)
                    with_bits.append(kw.get("bn_reduce") is not None and len(kw["bn_reduce"]) > 7)
                    with_bn2.append(kw.get("bn_reduce") is not None and len(kw["bn_reduce"]) > 8)
                    return orig(*a, **kw)
                layers.ops.conv_dgrad, layers.ops.bn_bwd_dx2 = spy, spy_dx2
                try:
                    _, loss, _ = framework_step(net, x, y)
                finally:
                    layers.ops.conv_dgrad, layers.ops.bn_bwd_dx2 = orig, orig_dx2
                res[(compact, relu_bits, bn2)] = (loss, grads_by_name(net), sum(made), sum(with_bits), sum(with_bn2), len(dx2))
            base = res[(True, True, False)]
            assert base[2] == 3 and res[(False, True, False)][2] == 0     # layer2.0, layer3.0, layer4.0 took a compact addend
            assert base[3] == 15 and res[(True, False, False)][3] == 0    # conv1 of every block after a residual ReLU read the bits
            for other in ((False, True, False), (True, False, False), (True, True, "own_apply")):
                assert res[other][0] == base[0]
                for k, gk in base[1].items():
                    assert torch.equal(res[other][1][k], gk), (other, k)
            # the downsample branches' BatchNorm reductions in the same epilogues (layer1.1 .. layer4.1 conv1): other
            # summation order than bn_bwd_reduce's, so to rounding (see the docstring for what that means here)
            fused2 = res[(True, True, True)]
            assert fused2[4] == 4 and base[4] == 0
            two_dx = res[(True, True, "two_dx")]
            assert fused2[5] == 4 and two_dx[5] == 0 and two_dx[4] == 4 and two_dx[0] == fused2[0]
            for k, gk in fused2[1].items():
                assert torch.equal(two_dx[1][k], gk), ("two_dx", k)
            assert abs(fused2[0] - base[0]) < 1e-4 * max(1.0, abs(base[0]))
            ga = torch.cat([fused2[1][k].double().flatten() for k in base[1]])
            gb = torch.cat([g.double().flatten() for g in base[1].values()])
            assert 1 - torch.dot(ga, gb).item() / (ga.norm().item() * gb.norm().item()) < 2e-3
            for k in base[1]:
                if "downsample.1" in k:       # the BatchNorm parameters whose gradients ARE those sums
                    torch.testing.assert_close(fused2[1][k], base[1][k], rtol=2e-3, atol=1e-4 * gmax)
        finally:
            layers._COMPACT_DOWN, layers._RELU_BITS, layers._FUSE_BN2, layers._DEFER_DOWN_BN, layers._BN_DX2 = keep
    finally:
        lib.dpa_x6s_knob(1, 1)
        set_fp32_mode(prev)


# (N, H, W, C = channels of dx, K = channels of dy, k, stride, pad) of the dgrad that TAKES the compact addend
COMPACT_SHAPES = [
    (4, 28, 28, 256, 128, 1, 1, 0),     # bottleneck conv1: 1x1, stride 1 (the division-free row -> pixel path)
    (3, 30, 30, 64, 128, 3, 2, 1),      # basic-block conv1: 3x3, stride 2, four parity classes
    (2, 17, 13, 128, 64, 3, 1, 1),      # odd map sizes: the compact tensor is ceil(H / 2) x ceil(W / 2)
]


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("shape", COMPACT_SHAPES)
def test_conv_dgrad_compact_addend(shape, fused, x6p_variant):
    """conv_dgrad(addend2=G): the compact gradient of a 1x1 / stride-2 convolution over the same input, added to the pixels
    with even coordinates in the epilogue of the pre-split fp32 kernel - against the dense addend it replaces (zeros with G
    at the even pixels): same bits, with and without the fused BatchNorm-backward reduction; a launch that cannot take
    it raises before anything runs."""
    from distributedpytorch_amd.ops import hip_conv, native
    from distributedpytorch_amd.ops.native import set_fp32_mode
    N, H, W, C, K, k, s, pd = shape
    g = torch.Generator(device="cuda").manual_seed(31)
    Ho, Wo = hip_conv.out_size(H, k, s, pd), hip_conv.out_size(W, k, s, pd)
    dy = torch.randn(N, Ho, Wo, K, device="cuda", generator=g)
    wt = torch.randn(C, k, k, K, device="cuda", generator=g) * (k * k * K) ** -0.5
    G = torch.randn(N, (H + 1) // 2, (W + 1) // 2, C, device="cuda", generator=g)
    z = torch.randn(N, H, W, C, device="cuda", generator=g)
    y = torch.randn(N, H, W, C, device="cuda", generator=g)
    mean = torch.randn(C, device="cuda", generator=g) * 0.3
    invstd = torch.rand(C, device="cuda", generator=g) + 0.5
    lib = native.hip_lib()
    prev, prev_tiles = set_fp32_mode("x6"), lib.dpa_x6p_min_tiles(1)
    try:
        wt3 = hip_conv.split3_planes(wt, torch.empty(3 * wt.numel(), device="cuda", dtype=torch.bfloat16))
        rows = hip_conv.bn_reduce_rows(N * H * W, (s, s))

        def run(**kw):
            dx = torch.full((N, H, W, C), float("nan"), device="cuda")
            slab = torch.full((rows, 2, C), float("nan"), device="cuda")
            if "dense" in kw:
                dx.copy_(kw["dense"])
                kw = dict(addend=dx)
            used = hip_conv.conv_dgrad(dy, wt, dx, (s, s), (pd, pd), wt3=wt3,
                                       bn_reduce=(z, y, mean, invstd, None, None, slab) if fused else None, **kw)
            return dx, slab[:used].clone()
        dense = torch.zeros(N, H, W, C, device="cuda")
        dense[:, ::2, ::2, :] = G
        want, want_slab = run(dense=dense)
        got, got_slab = run(addend2=G)
        assert torch.equal(got, want) and torch.equal(got_slab, want_slab)
        # a launch that stays on the general kernel (exact fp32 arithmetic) must refuse, and must not have touched dx
        set_fp32_mode("exact")
        dx = torch.full((N, H, W, C), 7.0, device="cuda")
        with pytest.raises(hip_conv.CompactAddendUnsupported):
            hip_conv.conv_dgrad(dy, wt, dx, (s, s), (pd, pd), addend2=G)
        torch.cuda.synchronize()
        assert (dx == 7.0).all()
    finally:
        lib.dpa_x6p_min_tiles(prev_tiles)
        set_fp32_mode(prev)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "fp32"])
@pytest.mark.parametrize("shape", [(32, 56, 64, 256, 1, 1, 0), (16, 224, 8, 64, 7, 2, 3)], ids=["1x1_64to256", "stem"])
def test_fused_statistics_are_the_same_from_launch_to_launch(shape, dtype, fp32_mode):
    """The BatchNorm statistics a convolution epilogue writes must not depend on timing.  Regression test: the bf16
    register-staged instantiations of conv_igemm.hip wrote a wrong sum of squares for a few 16-column groups of a few of
    the ~1500 tiles of such a launch, differently every time (a ds_bpermute issued right behind the masked ds_write2 of
    the previous column group overwrote that write's second data register) - invisible in the output and in any
    tolerance-based check, visible as bf16 training that was not reproducible at ImageNet sizes.
    scripts/diag/stats_race.py, scripts/diag/bf16_determinism.py."""
    from distributedpytorch_amd.ops import hip_conv
    if dtype == torch.bfloat16 and fp32_mode == "exact":
        pytest.skip("the fp32 arithmetic mode does not affect the bf16 kernels")
    n, h, c, k, ks, st, pd = shape
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(n, h, h, c, device="cuda", generator=g).to(dtype)
    w = (torch.randn(k, ks, ks, c, device="cuda", generator=g) * (ks * ks * c) ** -0.5).to(dtype)
    w3 = None
    if dtype == torch.float32 and fp32_mode == "x6" and c % 32 == 0:
        w3 = hip_conv.split3_planes(w, torch.empty(3 * w.numel(), device="cuda", dtype=torch.bfloat16))
    ho = hip_conv.out_size(h, ks, st, pd)
    rows = hip_conv.stats_rows(n * ho * ho, k, dtype)
    first = None
    for _ in range(12):
        y = torch.empty(n, ho, ho, k, device="cuda", dtype=dtype)
        slab = torch.zeros(rows, 2, k, device="cuda")
        hip_conv.conv_fprop(x, w, y, (st, st), (pd, pd), stats=slab, w3=w3)
        if first is None:
            first = (y, slab)
        else:
            assert torch.equal(y, first[0])
            assert torch.equal(slab, first[1]), (slab - first[1]).nonzero()[:4].tolist()
