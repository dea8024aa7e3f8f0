"""GPU parity tests, op level: every C-ABI kernel against a plain fp32 reference of the same op
(torch on the same device and/or the CPU oracle), at small sizes.  Stated tolerances are relative to
the reference's max magnitude; integer/index outputs must be exact."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import golden, rnd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    import evc_amd  # noqa: F401
    from evc_amd import lib
    lib.hip_lib()   # raises if libevc_hip.so is missing or the device is not gfx950: no silent fallback
    return lib


@pytest.fixture(params=[0, 1, 2], ids=["f32mfma", "bf16x6", "f16x3"])
def arith(request):
    """All three multiplication schemes of the convolution (include/evc_hip.h EVC_ARITH_*), same tolerances (the
    test inputs are O(1), which is what EVC_ARITH_F16X3 requires of its operands)."""
    return request.param


@pytest.fixture(params=[1, 2], ids=["bf16x6", "f16x3"])
def split_arith(request):
    return request.param


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def silu_affine(x, a, s):
    return F.silu(x * a[:, :, None, None] + s[:, :, None, None])


@pytest.mark.parametrize("B,H,W,Ci,Co,K", [(2, 16, 16, 192, 192, 3), (1, 5, 7, 32, 48, 3), (2, 8, 8, 64, 15, 3),
                                            (3, 8, 8, 96, 128, 1), (1, 9, 6, 32, 64, 5), (1, 1, 7, 192, 768, 1)])
def test_conv_plain(L, arith, B, H, W, Ci, Co, K):
    x = rnd(1, B, Ci, H, W).cuda()
    w = (rnd(2, Co, Ci, K, K) / np.sqrt(Ci * K * K)).cuda()
    b = rnd(3, Co).cuda()
    ref = F.conv2d(x.cpu(), w.cpu(), b.cpu(), padding=K // 2)
    out = L.conv2d_nhwc(nhwc(x), L.conv_pack_weights(w, arith), Co, K, K, bias=b)
    assert out.shape == (B, H, W, Co)
    assert rel(nchw(out), ref) < 1e-5


@pytest.mark.parametrize("splits", [0, 1, 3, 7])
def test_conv_fused_everything(L, arith, splits):
    """two-source concat + GroupNorm affine + SiLU on load + bias + residual + rescale (+ split-K)."""
    B, H, W, C0, C1, Co = 2, 8, 8, 96, 64, 192
    x0, x1 = rnd(4, B, C0, H, W).cuda(), rnd(5, B, C1, H, W).cuda()
    a, s = (1 + 0.2 * rnd(6, B, C0 + C1)).cuda(), (0.3 * rnd(7, B, C0 + C1)).cuda()
    w = (rnd(8, Co, C0 + C1, 3, 3) / np.sqrt(9 * (C0 + C1))).cuda()
    b, res = rnd(9, Co).cuda(), rnd(10, B, Co, H, W).cuda()
    xin = silu_affine(torch.cat([x0, x1], 1).cpu(), a.cpu(), s.cpu())
    ref = (F.conv2d(xin, w.cpu(), b.cpu(), padding=1) + res.cpu()) * 0.70710678
    out = L.conv2d_nhwc(nhwc(x0), L.conv_pack_weights(w, arith), Co, 3, 3, bias=b, src1=nhwc(x1), coef=(a, s),
                        act_in=L.ACT_SILU, res=nhwc(res), out_scale=0.70710678, splits=splits)
    assert rel(nchw(out), ref) < 1e-5


def test_conv_relu_in_out_and_wide_output_row(L, arith):
    B, H, W, Ci, Co = 1, 8, 8, 32, 16
    x = rnd(11, B, Ci, H, W).cuda()
    w = (rnd(12, Co, Ci, 3, 3) / np.sqrt(9 * Ci)).cuda()
    ref = F.relu(F.conv2d(F.relu(x.cpu()), w.cpu(), None, padding=1))
    out = torch.full((B, H, W, 24), 7.0, device="cuda")
    L.conv2d_nhwc(nhwc(x), L.conv_pack_weights(w, arith), Co, 3, 3, act_in=L.ACT_RELU, act_out=L.ACT_RELU, out=out)
    assert rel(nchw(out[..., :Co]), ref) < 1e-5
    assert bool((out[..., Co:] == 7.0).all())   # columns beyond Co are never written


def test_conv_full_size_layer_vs_device_reference(L, arith):
    """3x3 192->192 at 128x128 (the dominant layer shape), B=2, against torch on the same device."""
    B, H, W, C = 2, 128, 128, 192
    x = rnd(13, B, C, H, W).cuda()
    w = (rnd(14, C, C, 3, 3) / np.sqrt(9 * C)).cuda()
    b = rnd(15, C).cuda()
    ref = F.conv2d(x, w, b, padding=1)
    out = L.conv2d_nhwc(nhwc(x), L.conv_pack_weights(w, arith), C, 3, 3, bias=b)
    assert rel(nchw(out), ref) < 2e-5


def test_conv_fused_epilogue_moments_match_separate_pass(L, arith):
    """per-channel {sum, sumsq} of the conv output, produced in the epilogue (full tiles) or by the fallback pass."""
    for (B, H, W, Ci, Co, fused, sp) in ((2, 16, 16, 64, 192, True, 1), (1, 8, 24, 32, 192, True, 1), (1, 8, 20, 32, 192, False, 1),
                                          (2, 16, 16, 32, 96, False, 1), (2, 8, 8, 64, 96, True, 4), (3, 8, 8, 128, 15, True, 3)):
        x = rnd(16, B, Ci, H, W).cuda()
        w = (rnd(17, Co, Ci, 3, 3) / np.sqrt(9 * Ci)).cuda()
        res = rnd(18, B, Co, H, W).cuda()
        out, st = L.conv2d_nhwc(nhwc(x), L.conv_pack_weights(w, arith), Co, 3, 3, res=nhwc(res), out_scale=0.7,
                                want_stats=True, splits=sp)
        assert (L.conv_fused_stats_splits(B, H, W, Ci, Co, 3, 3, splits=sp, arith=arith) > 0) == fused
        o = out.double().reshape(B, H * W, Co)
        ref = torch.stack([o.sum(1), (o * o).sum(1)], -1).float().cpu()          # (B, Co, 2)
        got = st.double().sum(1).float().cpu()
        assert rel(got, ref) < 1e-5
        # and they drive GroupNorm exactly like the stand-alone moments
        G = 32 if Co % 32 == 0 else 3
        ca, cs = L.gn_coeffs([st], H * W, G, 1e-5)
        ca2, cs2 = L.gn_coeffs([L.chan_stats(out)], H * W, G, 1e-5) if Co % 4 == 0 else (ca, cs)
        assert rel(ca, ca2) < 1e-5 and rel(cs, cs2) < 1e-4


@pytest.mark.parametrize("B,H,W,C0,C1,Co,splits", [
    (3, 8, 8, 32, 16, 192, 0),      # W = 8: a 128-pixel tile spans two images; M = 192 leaves a partial last tile
    (3, 8, 8, 32, 16, 192, 3),      # same through split-K (splits cover whole kernel rows)
    (2, 16, 16, 48, 0, 192, 2),     # W = 16: eight image rows per tile
    (2, 32, 32, 16, 32, 64, 0),     # W = 32, Co = 64 (one 64-wide N tile), concat crossing a chunk boundary
    (1, 64, 64, 16, 0, 128, 0),     # W = 64: two image rows per tile, 128-wide N tile
    (1, 128, 128, 16, 16, 192, 0),  # W = 128: one image row per tile
    (1, 4, 4, 16, 0, 192, 0),       # W = 4: 32 image rows per tile, M = 16 (mostly padding)
    (8, 128, 128, 16, 16, 64, 0),   # 512 tiles of 256 pixels: the 8-wave / 256-pixel form of the kernel
])
def test_conv3x3_row_reuse_shapes(L, split_arith, B, H, W, C0, C1, Co, splits):
    """3x3 convolutions on tiles made of whole image rows run on the row-reuse kernel (activation staged once per
    kernel row, taps read at shifted LDS offsets, zero halo pixels for the horizontal borders): every image width
    it supports, tiles spanning images, partial tiles, concat sources, GroupNorm+SiLU on load, split-K."""
    x0 = rnd(50, B, C0, H, W).cuda()
    x1 = rnd(51, B, C1, H, W).cuda() if C1 else None
    C = C0 + C1
    a, s = (1 + 0.2 * rnd(52, B, C)).cuda(), (0.3 * rnd(53, B, C)).cuda()
    w = (rnd(54, Co, C, 3, 3) / np.sqrt(9 * C)).cuda()
    b, res = rnd(55, Co).cuda(), rnd(56, B, Co, H, W).cuda()
    xin = torch.cat([x0, x1], 1).cpu() if C1 else x0.cpu()
    ref = (F.conv2d(silu_affine(xin, a.cpu(), s.cpu()), w.cpu(), b.cpu(), padding=1) + res.cpu()) * 0.5
    out = L.conv2d_nhwc(nhwc(x0), L.conv_pack_weights(w, split_arith), Co, 3, 3, bias=b,
                        src1=None if x1 is None else nhwc(x1), coef=(a, s), act_in=L.ACT_SILU, res=nhwc(res),
                        out_scale=0.5, splits=splits)
    assert rel(nchw(out), ref) < 1e-5
    # plain (no transform) input through the same kernel
    ref2 = F.conv2d(xin, w.cpu(), None, padding=1)
    out2 = L.conv2d_nhwc(nhwc(x0), L.conv_pack_weights(w, split_arith), Co, 3, 3,
                         src1=None if x1 is None else nhwc(x1), splits=splits)
    assert rel(nchw(out2), ref2) < 1e-5


@pytest.mark.parametrize("B,H,W,C0,C1,Co", [(9, 64, 64, 64, 32, 384),     # 576 workgroups: one round + 32 tiles x 2 n-tiles split 2 ways
                                             (5, 128, 128, 96, 0, 192),    # 640 workgroups: one round + 128 tiles split 2 ways
                                             (5, 128, 128, 48, 48, 128)])  # 128-wide N tile (two 32-column MFMA tiles per wave)
def test_conv3x3_k_split_tail(L, split_arith, B, H, W, C0, C1, Co):
    """Unsplit grids of 1.x rounds: the pixel tiles of the partial last round are split along K inside the same launch
    (conv_tile_cfg, option "tail_split"), their slabs combined by a second kernel over the tail's rows only.  Outputs,
    fused moments (two producers: epilogue and combine) and the untouched main tiles against torch fp32 and against
    the same launch with the tail switched off."""
    x0 = rnd(60, B, C0, H, W).cuda()
    x1 = rnd(61, B, C1, H, W).cuda() if C1 else None
    C = C0 + C1
    a, s = (1 + 0.2 * rnd(62, B, C)).cuda(), (0.3 * rnd(63, B, C)).cuda()
    w = (rnd(64, Co, C, 3, 3) / np.sqrt(9 * C)).cuda()
    b, res = rnd(65, Co).cuda(), rnd(66, B, Co, H, W).cuda()
    wp = L.conv_pack_weights(w, split_arith)
    kw = dict(bias=b, src1=None if x1 is None else nhwc(x1), coef=(a, s), act_in=L.ACT_SILU, res=nhwc(res), out_scale=0.5)
    assert L.conv_workspace_bytes(B, H, W, C, Co, 3, 3, arith=split_arith) > 0          # the tail is in use ...
    assert L.conv_workspace_bytes(B, H, W, C, Co, 3, 3, arith=split_arith) < B * H * W * Co * 4   # ... and only the tail has slabs
    out, st = L.conv2d_nhwc(nhwc(x0), wp, Co, 3, 3, want_stats=True, **kw)
    L.conv_set_option("tail_split", 0)
    try:
        base, st0 = L.conv2d_nhwc(nhwc(x0), wp, Co, 3, 3, want_stats=True, splits=1, **kw)
    finally:
        L.conv_set_option("tail_split", 1)
    assert rel(out, base) < 2e-6
    assert rel(st.double().sum(1).float(), st0.double().sum(1).float()) < 1e-5
    xin = torch.cat([x0, x1], 1).cpu() if C1 else x0.cpu()
    ref = (F.conv2d(silu_affine(xin, a.cpu(), s.cpu()), w.cpu(), b.cpu(), padding=1) + res.cpu()) * 0.5
    assert rel(nchw(out), ref) < 1e-5


@pytest.mark.parametrize("B,H,W,C0,C1,Co,K,splits", [
    (2, 32, 32, 32, 16, 192, 3, 2),     # row-reuse kernel, blockIdx.z splits
    (3, 8, 8, 64, 0, 192, 3, 4),        # M = 192: a partial last pixel tile
    (2, 16, 16, 96, 32, 384, 1, 3),     # 1x1 filter: the simple-schedule kernel, two channel tiles
    (9, 64, 64, 32, 0, 192, 3, 0),      # automatic choice on a mid-size grid
    (5, 128, 128, 16, 16, 192, 3, 0),   # 640 tiles: one round + a K-split TAIL
])
def test_conv_split_k_is_deterministic_and_matches_unsplit(L, arith, B, H, W, C0, C1, Co, K, splits):
    """Split-K launches (blockIdx.z splits and the K-split tail, partial sums to slabs + the combine kernel that adds them
    in split order): against torch, against the same convolution forced unsplit (same values up to summation order), fused
    moments, and bitwise run-to-run determinism."""
    x0 = rnd(50, B, C0, H, W).cuda()
    x1 = rnd(51, B, C1, H, W).cuda() if C1 else None
    C = C0 + C1
    a, s = (1 + 0.2 * rnd(52, B, C)).cuda(), (0.3 * rnd(53, B, C)).cuda()
    w = (rnd(54, Co, C, K, K) / np.sqrt(K * K * C)).cuda()
    b, res = rnd(55, Co).cuda(), rnd(56, B, Co, H, W).cuda()
    xin = torch.cat([x0, x1], 1) if C1 else x0
    ref = (F.conv2d(silu_affine(xin, a, s), w, b, padding=K // 2) + res) * 0.5
    wp = L.conv_pack_weights(w, arith)
    kw = dict(bias=b, src1=None if x1 is None else nhwc(x1), coef=(a, s), act_in=L.ACT_SILU, res=nhwc(res), out_scale=0.5,
              want_stats=True)
    out, st = L.conv2d_nhwc(nhwc(x0), wp, Co, K, K, splits=splits, **kw)
    assert rel(nchw(out), ref) < 2e-5
    o = out.double().reshape(B, H * W, Co)
    assert rel(st.double().sum(1).float(), torch.stack([o.sum(1), (o * o).sum(1)], -1).float()) < 1e-5
    for _ in range(3):
        again, st2 = L.conv2d_nhwc(nhwc(x0), wp, Co, K, K, splits=splits, **kw)
        assert torch.equal(again, out) and torch.equal(st2, st)
    L.conv_set_option("tail_split", 0)
    try:
        one, _ = L.conv2d_nhwc(nhwc(x0), wp, Co, K, K, splits=1, **kw)
    finally:
        L.conv_set_option("tail_split", 1)
    assert rel(out, one) < 2e-6
    with pytest.raises(L.EvcKernelError):
        L.conv_set_option("no_such_option", 1)


@pytest.mark.parametrize("B,H,W,C,Co,C2a,C2b,splits", [
    (9, 32, 32, 192, 192, 96, 32, 0),     # concat second operand (two sources); split-K chosen automatically
    (9, 32, 32, 64, 192, 48, 0, 3),       # fixed 3-way split-K: every split takes its share of the 1x1 chunks
    (9, 8, 8, 384, 384, 160, 0, 0),       # W = 8 (a tile spans two images), two channel tiles, partial last pixel tile
    (9, 16, 16, 192, 384, 64, 64, 0),     # W = 16, two channel tiles, two sources
    (5, 128, 128, 96, 192, 64, 0, 0),     # 640 tiles: one round + a K-split tail
    (9, 64, 64, 192, 192, 192, 192, 0),   # wide kernel, 144 tiles: two unequal K pieces, each with half of the 24 1x1 chunks
    (8, 64, 64, 64, 128, 16, 0, 0),       # unsplit, a single 1x1 chunk, 128-wide channel tile
])
def test_conv3x3_with_fused_1x1_operand(L, B, H, W, C, Co, C2a, C2b, splits):
    """A res-block's Conv_1 (3x3 on GroupNorm + SiLU of h) and its skip convolution Conv_2 (1x1 on the raw block input, which
    may be a concat) in ONE launch and one accumulator (evc_conv_args.x2_*; reference models/better/layerspp.py:603-624:
    (Conv_2(x) + Conv_1(act(h))) / sqrt(2)).  Against torch; against the two separate launches it replaces; with a second
    operand 1e4 times larger / smaller than the first (the accumulator rescaling between the operands is exact)."""
    assert L.conv_fused_1x1_supported(B, H, W, C, Co, L.ARITH_F16X3, splits)
    assert not L.conv_fused_1x1_supported(B, H, W, C, Co, L.ARITH_BF16X6, splits)
    assert not L.conv_fused_1x1_supported(1, 8, 8, C, Co, L.ARITH_F16X3)        # tiny grid: 64-pixel tiles, no row-reuse kernel
    h = nhwc(rnd(60, B, C, H, W).cuda())
    a, s = (1 + 0.2 * rnd(61, B, C)).cuda(), (0.3 * rnd(62, B, C)).cuda()
    w1 = (rnd(63, Co, C, 3, 3) / np.sqrt(9 * C)).cuda()
    b1, b2 = rnd(64, Co).cuda(), rnd(65, Co).cuda()
    C2 = C2a + C2b
    w2 = (rnd(66, Co, C2, 1, 1) / np.sqrt(C2)).cuda()
    wp1, wp2 = L.conv_pack_weights(w1, L.ARITH_F16X3), L.conv_pack_weights(w2, L.ARITH_F16X3)
    for mag in (1.0, 1e4, 1e-4):
        xa = nhwc(rnd(67, B, C2a, H, W).cuda()) * mag
        xb = nhwc(rnd(68, B, C2b, H, W).cuda()) * (2 * mag) if C2b else None
        bound = torch.zeros(1, dtype=torch.int32, device="cuda")
        L.gn_coeffs([L.chan_stats(xa)] + ([L.chan_stats(xb)] if C2b else []), H * W, 16, 1e-5, bound=bound)
        fused, st = L.conv2d_nhwc(h, wp1, Co, 3, 3, bias=b1 + b2, coef=(a, s), act_in=L.ACT_SILU, out_scale=0.70710678,
                                  splits=splits, want_stats=True, x2=(xa, xb, wp2, bound))
        xcat = torch.cat([xa, xb], 3) if C2b else xa
        ref = (F.conv2d(silu_affine(nchw(h), a, s).double(), w1.double(), b1.double(), padding=1) +
               F.conv2d(nchw(xcat).double(), w2.double(), b2.double())) * 0.70710678
        assert float((nchw(fused).double() - ref).abs().max() / ref.abs().max()) < 3e-6, mag
        o = fused.double().reshape(B, H * W, Co)
        assert rel(st.double().sum(1).float(), torch.stack([o.sum(1), (o * o).sum(1)], -1).float()) < 1e-5
        skip = L.conv2d_nhwc(xa, wp2, Co, 1, 1, bias=b2, src1=xb, in_bound=bound)           # the two launches it replaces
        two = L.conv2d_nhwc(h, wp1, Co, 3, 3, bias=b1, coef=(a, s), act_in=L.ACT_SILU, res=skip, out_scale=0.70710678,
                            splits=splits)
        assert rel(fused, two) < 3e-6, mag
        again, _ = L.conv2d_nhwc(h, wp1, Co, 3, 3, bias=b1 + b2, coef=(a, s), act_in=L.ACT_SILU, out_scale=0.70710678,
                                 splits=splits, want_stats=True, x2=(xa, xb, wp2, bound))
        assert torch.equal(again, fused)
    # a NaN in the second operand: its bound is the NaN pattern, the whole output is NaN (never finite garbage)
    xn = xa.clone()
    xn[0, 1, 1, 3] = float("nan")
    bound.zero_()
    L.gn_coeffs([L.chan_stats(xn)] + ([L.chan_stats(xb)] if C2b else []), H * W, 16, 1e-5, bound=bound)
    bad = L.conv2d_nhwc(h, wp1, Co, 3, 3, bias=b1 + b2, coef=(a, s), act_in=L.ACT_SILU, splits=splits, x2=(xn, xb, wp2, bound))
    assert not bool(torch.isfinite(bad).any())
    L.range_events(reset=True)
    # where the fused operand is not available the call is refused, never silently ignored
    with pytest.raises(L.EvcKernelError):
        L.conv2d_nhwc(h, L.conv_pack_weights(w1, L.ARITH_BF16X6), Co, 3, 3, bias=b1, x2=(xa, xb, wp2, bound))


@pytest.mark.parametrize("B,H,W,C0,C1,Co,mode,what", [
    (4, 128, 128, 32, 16, 192, "gn", "256 tiles = exactly one round, two sources"),
    (5, 128, 128, 192, 0, 192, "gn", "320 tiles: one round + a 4-way K-split tail (12 chunks)"),
    (5, 128, 128, 48, 0, 192, "plain", "320 tiles, 3 chunks: tail too short to split -> a second round of whole tiles"),
    (9, 64, 64, 96, 0, 384, "gn", "288 workgroups over two channel tiles: one round + a split tail"),
    (9, 64, 64, 64, 0, 192, "plain", "144 tiles, 4 chunks: below one round, unsplit"),
    (9, 64, 64, 128, 64, 192, "gn", "144 tiles, 12 chunks: below one round, two UNEQUAL K pieces per tile (option wide_cut) + combine"),
    (9, 32, 32, 128, 64, 384, "gn", "72 tiles, 12 chunks: below one round, uniform 3-way K split + combine"),
    (64, 16, 16, 64, 0, 384, "gn", "W = 16: sixteen image rows per tile"),
])
def test_conv_wide_kernel_against_torch_and_the_row_reuse_kernel(L, B, H, W, C0, C1, Co, mode, what):
    """conv_wide_kernel (round 4: 256 x 192 tile, one 4-wave workgroup per CU, weights from L2 straight into registers, nine taps
    from one staged chunk image) on every grid class its dispatch takes -- whole rounds, round + K-split tail, sub-round grids
    unsplit and uniformly split -- and every image width: against torch in fp64 (GroupNorm + SiLU on load or plain, bias,
    residual, 1 / sqrt(2)), its fused moments against the output, and against the row-reuse kernel the same call ran on before
    (dispatch option "wide256" = 0): two kernels, one result up to fp32 summation order."""
    x0 = rnd(70, B, C0, H, W).cuda()
    x1 = rnd(71, B, C1, H, W).cuda() if C1 else None
    C = C0 + C1
    a, s = (1 + 0.2 * rnd(72, B, C)).cuda(), (0.3 * rnd(73, B, C)).cuda()
    w = (rnd(74, Co, C, 3, 3) / np.sqrt(9 * C)).cuda()
    b, res = rnd(75, Co).cuda(), rnd(76, B, Co, H, W).cuda()
    wp = L.conv_pack_weights(w, L.ARITH_F16X3)
    xin = torch.cat([x0, x1], 1) if C1 else x0
    kw = dict(bias=b, src1=None if x1 is None else nhwc(x1), res=nhwc(res), out_scale=0.70710678, want_stats=True)
    if mode == "gn":
        kw.update(coef=(a, s), act_in=L.ACT_SILU)
        pre = silu_affine(xin, a, s)
    else:
        pre = xin
    ref = (F.conv2d(pre.double(), w.double(), b.double(), padding=1) + res.double()) * 0.70710678

    def run():
        prof = []
        L.CONV_PROFILE = prof
        try:
            r = L.conv2d_nhwc(nhwc(x0), wp, Co, 3, 3, **kw)
        finally:
            L.CONV_PROFILE = None
        return r, prof[0]["kernel"]

    def moments(t):
        o = t.double().reshape(B, H * W, Co)
        return torch.stack([o.sum(1), (o * o).sum(1)], -1).float()
    (out, st), kernel = run()
    assert kernel.startswith("conv_wide_kernel<"), (what, kernel)
    err = float((nchw(out).double() - ref).abs().max() / ref.abs().max())
    assert err < 3e-6, (what, err)
    assert rel(st.double().sum(1).float(), moments(out)) < 1e-5, what
    again, _ = L.conv2d_nhwc(nhwc(x0), wp, Co, 3, 3, **kw)
    assert torch.equal(again, out), what                                   # deterministic
    try:
        L.conv_set_option("wide256", 0)
        (old, st_old), kernel = run()
    finally:
        L.conv_set_option("wide256", 1)
    assert not kernel.startswith("conv_wide_kernel<"), (what, kernel)
    assert rel(out, old) < 2e-6, what
    assert rel(st_old.double().sum(1).float(), moments(old)) < 1e-5, what


def test_conv_bf16x6_is_not_less_accurate_than_f32_mfma(L):
    """The precision claim of EVC_ARITH_BF16X6 and EVC_ARITH_F16X3, on the real kernels: against an fp64 reference
    their error is no larger than that of the exact-product f32 MFMA path (all accumulate in fp32), for the dominant
    layer shape with GroupNorm+SiLU on load and for a long-K split-K layer.  Same criterion for both."""
    for (B, H, W, Ci, Co, seed) in ((1, 32, 32, 192, 192, 40), (2, 8, 8, 1536, 768, 41)):
        x = rnd(seed, B, Ci, H, W).cuda()
        a, s = (1 + 0.2 * rnd(seed + 1, B, Ci)).cuda(), (0.3 * rnd(seed + 2, B, Ci)).cuda()
        w = (rnd(seed + 3, Co, Ci, 3, 3) / np.sqrt(9 * Ci)).cuda()
        xin = F.silu(x.double() * a.double()[:, :, None, None] + s.double()[:, :, None, None])
        ref = F.conv2d(xin, w.double(), None, padding=1)
        err = []
        for ar in (L.ARITH_F32, L.ARITH_BF16X6, L.ARITH_F16X3):
            out = L.conv2d_nhwc(nhwc(x), L.conv_pack_weights(w, ar), Co, 3, 3, coef=(a, s), act_in=L.ACT_SILU)
            e = (nchw(out).double() - ref).abs()
            err.append((float(e.max() / ref.abs().max()), float(e.pow(2).mean().sqrt() / ref.abs().max())))
        (mx32, rms32), (mx6, rms6), (mx3, rms3) = err
        # the on-load SiLU uses the hardware exp/rcp (~1e-7 relative), common to all paths
        assert mx32 < 5e-6 and mx6 < 5e-6 and mx3 < 5e-6, err
        assert rms6 <= 1.25 * rms32 and mx6 <= 2.0 * mx32, err
        assert rms3 <= 1.25 * rms32 and mx3 <= 2.0 * mx32, err


def test_conv_f16x3_scaling_and_loud_overflow(L):
    """EVC_ARITH_F16X3 range handling: the per-tensor weight scale makes tiny and huge weights equally accurate (the
    inverse is applied to the accumulator), small activations keep their precision (absolute error of the split is
    2^-25 / 8).  Nothing is clamped: an activation beyond fp16's range (|x| > 65504 / 8), a NaN and an infinity all come
    out NON-FINITE at the pixels they reach -- never as silently saturated finite numbers -- and leave the rest alone."""
    B, H, W, Ci, Co = 1, 16, 16, 64, 192
    x = rnd(70, B, Ci, H, W).cuda()
    w0 = rnd(71, Co, Ci, 3, 3) / np.sqrt(9 * Ci)
    for wscale, xscale in ((1.0, 1.0), (1e-6, 1.0), (3e4, 1.0), (1.0, 0.01)):
        w = (w0 * wscale).cuda()
        ref = F.conv2d((x * xscale).double(), w.double(), None, padding=1)
        out = L.conv2d_nhwc(nhwc(x * xscale), L.conv_pack_weights(w, L.ARITH_F16X3), Co, 3, 3)
        assert float((nchw(out).double() - ref).abs().max() / ref.abs().max()) < 2e-6, (wscale, xscale)
    wp = L.conv_pack_weights(w0.cuda(), L.ARITH_F16X3)
    far = torch.ones(H, W, dtype=torch.bool); far[4:7, 4:7] = False     # pixels the bad element does not reach
    ref = F.conv2d(x.cpu(), w0, None, padding=1)
    for bad in (1e6, float("nan"), float("inf"), -float("inf")):         # 1e6: far beyond 65504 / 8
        xb = x.clone()
        xb[0, 3, 5, 5] = bad
        out = nchw(L.conv2d_nhwc(nhwc(xb), wp, Co, 3, 3))[0].cpu()
        assert not bool(torch.isfinite(out[:, 4:7, 4:7]).any()), bad     # every output the element feeds is non-finite
        assert bool(torch.isfinite(out[:, far]).all()) and rel(out[:, far], ref[0][:, far]) < 1e-5, bad
    # the largest magnitude that still fits is exact business as usual
    xb = x.clone()
    xb[0, 3, 5, 5] = 8000.0
    out = nchw(L.conv2d_nhwc(nhwc(xb), wp, Co, 3, 3))
    refb = F.conv2d(xb.double(), w0.cuda().double(), None, padding=1)
    assert float((out.double() - refb).abs().max() / refb.abs().max()) < 2e-6


def test_range_events_from_the_coefficient_kernels(L):
    """include/evc_hip.h EVC_RANGE_*: the kernels that see every tensor's moments report (sticky device word) a tensor
    that holds a NaN / inf and a GroupNorm-ed operand that may leave fp16's range; in-range finite tensors report nothing;
    a non-finite tensor turns its element bound into NaN, so the convolution reading it through ``in_bound`` gives NaN."""
    L.range_events(reset=True)
    B, H, W, C = 2, 16, 16, 64
    x = nhwc(rnd(75, B, C, H, W).cuda())
    st = L.chan_stats(x)
    bound = torch.zeros(1, dtype=torch.int32, device="cuda")
    table = torch.zeros(1, 2 * C, device="cuda")
    row = torch.zeros(B, dtype=torch.int32, device="cuda")
    L.gn_coeffs([st], H * W, 32, 1e-5, mode=2, ss=table, row=row, bound=bound)
    assert L.range_events() == 0
    # AdaGN scale row of 3000: |gn(x) * 3001| can pass 8188 -> reported (a sufficient condition: "may")
    table[0, :C] = 3000.0
    L.gn_coeffs([st], H * W, 32, 1e-5, mode=2, ss=table, row=row)
    assert L.range_events(reset=True) == L.RANGE_F16_OPERAND
    assert L.range_events() == 0
    # a NaN in the tensor: moments not finite -> NONFINITE, bound word = NaN pattern, the 1x1 convolution outputs NaN
    xn = x.clone()
    xn[1, 3, 3, 7] = float("nan")
    bound.zero_()
    ca, cs = L.gn_coeffs([L.chan_stats(xn)], H * W, 32, 1e-5, bound=bound)
    assert L.range_events() & L.RANGE_NONFINITE
    assert not bool(torch.isfinite(bound.view(torch.float32)).all())
    assert not bool(torch.isfinite(ca[1]).all()) and bool(torch.isfinite(ca[0]).all())     # only the sample that holds it
    w = (rnd(76, 192, C, 1, 1) / 8).cuda()
    out = L.conv2d_nhwc(xn, L.conv_pack_weights(w, L.ARITH_F16X3), 192, 1, 1, in_bound=bound)
    assert not bool(torch.isfinite(out).any())
    # ... and through the GroupNorm-on-load path (NaN coefficients): the whole sample is NaN, the other one is untouched
    w3 = (rnd(77, 192, C, 3, 3) / 24).cuda()
    out = L.conv2d_nhwc(xn, L.conv_pack_weights(w3, L.ARITH_F16X3), 192, 3, 3, coef=(ca, cs), act_in=L.ACT_SILU)
    assert not bool(torch.isfinite(out[1]).any()) and bool(torch.isfinite(out[0]).all())
    L.range_events(reset=True)
    qkv = L.chan_stats(nhwc(rnd(78, B, 96, H, W).cuda()))
    b3 = torch.zeros(3, dtype=torch.int32, device="cuda")
    L.moments_bound(qkv, 0, 32, b3)
    assert L.range_events() == 0 and bool((b3.view(torch.float32) > 0).all())
    qkv[0, 0, 40, 1] = float("inf")
    b3.zero_()
    L.moments_bound(qkv, 0, 32, b3)
    f = b3.view(torch.float32)
    assert L.range_events(reset=True) == L.RANGE_NONFINITE and bool(torch.isnan(f[1])) and bool(torch.isfinite(f[[0, 2]]).all())


@pytest.mark.parametrize("K,H", [(1, 32), (3, 16)])
def test_conv_f16x3_on_raw_input_with_moment_bound(L, K, H):
    """EVC_ARITH_F16X3 on an input no GroupNorm has normalised (1x1 skip convolutions, NIN output projections): the
    element bound comes from the tensor's per-channel moments (evc_gn_coeffs_bound_f32 / evc_moments_bound_f32), the
    kernel scales by the matching power of two.  Inputs of very different magnitudes -- 1e-5 ... 1e+5, far outside
    what the unscaled fp16 split could represent -- reach the accuracy of the bf16x6 path."""
    B, W, C0, C1, Co = 2, H, 64, 32, 192
    w = (rnd(80, Co, C0 + C1, K, K) / np.sqrt((C0 + C1) * K * K)).cuda()
    wp = L.conv_pack_weights(w, L.ARITH_F16X3)
    for mag in (1.0, 1e-5, 1e5):
        x0, x1 = (rnd(81, B, C0, H, W) * mag).cuda(), (rnd(82, B, C1, H, W) * mag * 7).cuda()
        x0[1, 5, 3, 3] = 40 * mag                       # an outlier well inside the bound
        n0, n1 = nhwc(x0), nhwc(x1)
        bound = torch.zeros(1, dtype=torch.int32, device="cuda")
        L.gn_coeffs([L.chan_stats(n0), L.chan_stats(n1)], H * W, 32, 1e-5, bound=bound)
        bval = float(bound.view(torch.float32).sqrt())
        assert bval >= float(torch.cat([x0, x1], 1).abs().max()) > 0          # it IS a bound on every element
        ref = F.conv2d(torch.cat([x0, x1], 1).double(), w.double(), None, padding=K // 2)
        out = L.conv2d_nhwc(n0, wp, Co, K, K, src1=n1, in_bound=bound)
        e16 = float((nchw(out).double() - ref).abs().max() / ref.abs().max())
        out6 = L.conv2d_nhwc(n0, L.conv_pack_weights(w, L.ARITH_BF16X6), Co, K, K, src1=n1)
        e6 = float((nchw(out6).double() - ref).abs().max() / ref.abs().max())
        assert e16 < 2e-6 and e16 <= 2.0 * e6 + 2e-7, (mag, e16, e6)
    # the bound of a channel range of one moments tensor (the value third of a q|k|v tensor)
    qkv = nhwc(torch.cat([x0 * 1e3, x0, x1[:, :32] * 3], 1))          # channels 0..63 huge, 64..159 moderate
    b2 = torch.zeros(1, dtype=torch.int32, device="cuda")
    L.moments_bound(L.chan_stats(qkv), 64, 96, b2)
    b3 = torch.zeros(2, dtype=torch.int32, device="cuda")           # two consecutive ranges in one launch
    L.moments_bound(L.chan_stats(qkv), 0, 64, b3)
    assert float(b3[:1].view(torch.float32).sqrt()) >= float(qkv[..., :64].abs().max())
    assert float(b3[1:].view(torch.float32).sqrt()) >= float(qkv[..., 64:128].abs().max())
    assert float(b2.view(torch.float32).sqrt()) >= float(qkv[..., 64:].abs().max())
    assert float(b2.view(torch.float32).sqrt()) < float(qkv[..., :64].abs().max())      # the other channels do not count
    with pytest.raises(L.EvcKernelError):        # only the fp16 split scales its input
        L.conv2d_nhwc(n0, L.conv_pack_weights(w, L.ARITH_BF16X6), Co, K, K, src1=n1, in_bound=bound)


def test_conv_rejects_bad_arguments(L):
    x = torch.zeros(1, 4, 4, 24, device="cuda")   # 24 channels: not a multiple of 16
    w = torch.zeros(64 * 2 * 16, device="cuda")
    with pytest.raises(L.EvcKernelError):
        L.conv2d_nhwc(x, w, 64, 3, 3)
    with pytest.raises(L.EvcKernelError):
        L.conv_pack_weights(torch.zeros(64, 32, 3, 3, device="cuda"), arith=7)
    with pytest.raises(ValueError):
        import os
        old = os.environ.get("EVC_CONV_ARITH")
        os.environ["EVC_CONV_ARITH"] = "fp8"
        try:
            L.default_arith()
        finally:
            os.environ.pop("EVC_CONV_ARITH") if old is None else os.environ.__setitem__("EVC_CONV_ARITH", old)


@pytest.mark.parametrize("B,H,W,C,G", [(2, 16, 16, 192, 32), (3, 8, 8, 1344, 32), (1, 32, 32, 32, 8), (2, 4, 4, 160, 32)])
def test_groupnorm_via_moments_and_coeffs(L, B, H, W, C, G):
    x = (rnd(20, B, C, H, W) * 1.7 + 0.4).cuda()
    gamma, beta = (1 + 0.1 * rnd(21, C)).cuda(), (0.1 * rnd(22, C)).cuda()
    xh = nhwc(x)
    part = L.chan_stats(xh)
    ca, cs = L.gn_coeffs([part], H * W, G, 1e-6, mode=1, gamma=gamma, beta=beta)
    ref = F.group_norm(x.cpu(), G, gamma.cpu(), beta.cpu(), 1e-6)
    assert rel(nchw(L.affine_act(xh, (ca, cs), L.ACT_NONE)), ref) < 1e-5
    # AdaGN + SiLU with a two-row table and per-sample rows
    ss = (0.2 * rnd(23, 2, 2 * C + 8)).cuda()
    row = torch.tensor([1, 0, 1][:B], dtype=torch.int32, device="cuda")
    ca, cs = L.gn_coeffs([part], H * W, G, 1e-5, mode=2, ss=ss[:, 8:], row=row)
    sc = ss[row.long(), 8:8 + C].cpu()
    sh = ss[row.long(), 8 + C:].cpu()
    ref = F.silu(F.group_norm(x.cpu(), G, None, None, 1e-5) * (1 + sc[:, :, None, None]) + sh[:, :, None, None])
    assert rel(nchw(L.affine_act(xh, (ca, cs), L.ACT_SILU)), ref) < 1e-5


def test_groupnorm_groups_straddling_a_concat(L):
    """cat[h (96 ch), skip (64 ch)] -> 160 channels in 32 groups of 5: groups cross the tensor boundary."""
    B, H, W = 2, 8, 8
    h, sk = rnd(24, B, 96, H, W).cuda(), (2 * rnd(25, B, 64, H, W) - 1).cuda()
    ca, cs = L.gn_coeffs([L.chan_stats(nhwc(h)), L.chan_stats(nhwc(sk))], H * W, 32, 1e-5)
    cat = torch.cat([h, sk], 1)
    got = cat.cpu() * ca.cpu()[:, :, None, None] + cs.cpu()[:, :, None, None]
    assert rel(got, F.group_norm(cat.cpu(), 32, None, None, 1e-5)) < 1e-5


def test_upfirdn2d_dropin_against_reference_goldens(L):
    g = golden("fir")
    x = torch.from_numpy(g["x"]).cuda()
    k = np.outer([1, 3, 3, 1], [1, 3, 3, 1]).astype(np.float32)
    k /= k.sum()
    up = L.upfirdn2d_nchw(x, k * 4, up=2, pad=(2, 1))
    down = L.upfirdn2d_nchw(x, k, down=2, pad=(1, 1))
    assert up.shape == g["up"].shape and down.shape == g["down"].shape
    np.testing.assert_allclose(up.cpu().numpy(), g["up"], atol=2e-6)
    np.testing.assert_allclose(down.cpu().numpy(), g["down"], atol=2e-6)
    g2 = golden("upfirdn2d_generic")
    x2, k2 = torch.from_numpy(g2["x"]).cuda(), g2["k"]
    np.testing.assert_allclose(L.upfirdn2d_nchw(x2, k2, up=2, down=1, pad=(1, 0)).cpu().numpy(), g2["up2_pad10"], atol=3e-6)
    np.testing.assert_allclose(L.upfirdn2d_nchw(x2, k2, up=1, down=3, pad=(2, 1)).cpu().numpy(), g2["down3_pad21"], atol=3e-6)
    np.testing.assert_allclose(L.upfirdn2d_nchw(x2, k2, up=3, down=2, pad=(2, 2)).cpu().numpy(), g2["up3_down2_pad22"], atol=3e-6)


def test_upfirdn2d_nhwc_with_fused_adagn_silu(L):
    from oracle import upfirdn2d as O
    B, C, H, W = 2, 32, 12, 10
    x = rnd(30, B, C, H, W).cuda()
    a, s = (1 + 0.2 * rnd(31, B, C)).cuda(), (0.3 * rnd(32, B, C)).cuda()
    k = O.setup_kernel([1, 3, 3, 1])
    act = silu_affine(x.cpu(), a.cpu(), s.cpu()).numpy()
    got = L.upfirdn2d_nhwc(nhwc(x), k * 4, 2, 1, (2, 1), coef=(a, s), act=L.ACT_SILU)
    np.testing.assert_allclose(nchw(got).cpu().numpy(), O.upsample_2d(act), atol=3e-6)
    got = L.upfirdn2d_nhwc(nhwc(x), k, 1, 2, (1, 1), coef=(a, s), act=L.ACT_SILU)
    np.testing.assert_allclose(nchw(got).cpu().numpy(), O.downsample_2d(act), atol=3e-6)
    got = L.upfirdn2d_nhwc(nhwc(x), k, 1, 2, (1, 1))
    np.testing.assert_allclose(nchw(got).cpu().numpy(), O.downsample_2d(x.cpu().numpy()), atol=3e-6)
    # zero insertion used by the ELIC transposed convolutions: k = [[1]], up 2, pad (0, 0) -> 2H x 2W
    got = L.upfirdn2d_nhwc(nhwc(x), np.ones((1, 1), np.float32), 2, 1, (0, 0))
    z = torch.zeros(B, C, 2 * H, 2 * W)
    z[:, :, ::2, ::2] = x.cpu()
    assert torch.equal(nchw(got).cpu(), z)


@pytest.mark.parametrize("B,C,H,W", [(2, 32, 16, 12), (1, 192, 8, 8), (3, 48, 4, 4), (1, 16, 32, 64)])
def test_upfirdn2d_nhwc_fast_paths_with_an_asymmetric_kernel(L, B, C, H, W):
    """the specialised up-2 / down-2 kernels (4x4 taps, blocks of outputs per thread) against the oracle with a
    NON-separable, asymmetric kernel (catches flipped or transposed taps), with and without AdaGN+SiLU on load,
    including sizes where every block touches an image border."""
    from oracle import upfirdn2d as O
    x = rnd(33, B, C, H, W).cuda()
    a, s = (1 + 0.2 * rnd(34, B, C)).cuda(), (0.3 * rnd(35, B, C)).cuda()
    k = (rnd(36, 4, 4).abs() + 0.1).numpy().astype(np.float32)
    act = silu_affine(x.cpu(), a.cpu(), s.cpu()).numpy()
    for src, kw in ((act, dict(coef=(a, s), act=L.ACT_SILU)), (x.cpu().numpy(), {})):
        got = L.upfirdn2d_nhwc(nhwc(x), k, 2, 1, (2, 1), **kw)
        np.testing.assert_allclose(nchw(got).cpu().numpy(), O.upfirdn2d(src, k, up=2, down=1, pad=(2, 1)), atol=5e-6)
        got = L.upfirdn2d_nhwc(nhwc(x), k, 1, 2, (1, 1), **kw)
        np.testing.assert_allclose(nchw(got).cpu().numpy(), O.upfirdn2d(src, k, up=1, down=2, pad=(1, 1)), atol=5e-6)


def qkv_bounds(L, qkv, C):
    """Element bounds of the q | k | v thirds, as ScoreNet derives them from the projection's moments."""
    B, N, _ = qkv.shape
    b = torch.zeros(3, dtype=torch.int32, device="cuda")
    L.moments_bound(L.chan_stats(qkv.view(B, 1, N, 3 * C)), 0, C, b)
    return b


@pytest.fixture(params=[False, True], ids=["f32mfma", "f16x3"])
def attn_f16(request):
    return request.param


@pytest.mark.parametrize("B,heads,N,D", [(2, 2, 1024, 192), (1, 4, 64, 192), (2, 3, 256, 192), (2, 2, 96, 32), (1, 1, 64, 64),
                                         (2, 1, 256, 128), (1, 2, 64, 128), (2, 1, 256, 256), (1, 1, 100, 256)])
def test_attention(L, attn_f16, B, heads, N, D):
    C = heads * D
    qkv = rnd(40, B, N, 3 * C).cuda()
    out = L.attention(qkv, C, heads, bounds=qkv_bounds(L, qkv, C) if attn_f16 else None)
    q, k, v = [t.reshape(B, N, heads, D).permute(0, 2, 1, 3) for t in qkv.cpu().split(C, dim=2)]
    w = torch.softmax(torch.einsum("bhqd,bhkd->bhqk", q, k) * (D ** -0.5), dim=-1)
    ref = torch.einsum("bhqk,bhkd->bhqd", w, v).permute(0, 2, 1, 3).reshape(B, N, C)
    assert rel(out, ref) < 1e-5


def test_attention_peaked_scores_exercise_the_rescale(L, attn_f16):
    """large logits whose maximum moves to later key tiles force the online-softmax rescale branch."""
    B, heads, N, D = 1, 1, 128, 32
    C = D
    q = rnd(41, B, N, C)
    k = rnd(42, B, N, C) * torch.linspace(0.5, 6.0, N)[None, :, None]   # later keys dominate
    v = rnd(43, B, N, C)
    qkv = torch.cat([q, k, v], 2).cuda()
    out = L.attention(qkv, C, heads, bounds=qkv_bounds(L, qkv, C) if attn_f16 else None)
    w = torch.softmax(torch.einsum("bqd,bkd->bqk", q.double(), k.double()) * (D ** -0.5), dim=-1)
    ref = torch.einsum("bqk,bkd->bqd", w, v.double()).float()
    assert rel(out, ref) < 2e-5


def test_attention_key_split_matches_single_pass(L):
    """evc_attention_ws_f32 splits the key range over workgroups when a launch would leave SIMDs idle and merges the
    partial softmax states; it must agree with the single-pass evc_attention_f32 (and with fp64), including a ragged
    last key tile and key parts whose maxima differ by a lot."""
    import ctypes
    lib = L.hip_lib()
    for (B, heads, N, D) in ((1, 2, 200, 192), (2, 1, 1024, 64), (9, 2, 1024, 192)):
        C = heads * D
        q = rnd(44, B, N, C)
        k = rnd(45, B, N, C) * torch.linspace(0.3, 4.0, N)[None, :, None]
        v = rnd(46, B, N, C)
        qkv = torch.cat([q, k, v], 2).cuda()
        nbytes = lib.evc_attention_workspace_bytes(B, heads, N, D)
        assert nbytes > 0                                      # these shapes do split
        ws = torch.empty(nbytes // 4, device="cuda")
        out_ws, out_1 = torch.empty(B, N, C, device="cuda"), torch.empty(B, N, C, device="cuda")
        base = qkv.data_ptr()
        args = (ctypes.c_void_p(base), ctypes.c_void_p(base + 4 * C), ctypes.c_void_p(base + 8 * C), 3 * C)
        assert lib.evc_attention_ws_f32(*args, L.fptr(out_ws), C, B, heads, N, D, D ** -0.5, L.ptr(ws), L.stream_ptr()) == 0
        assert lib.evc_attention_f32(*args, L.fptr(out_1), C, B, heads, N, D, D ** -0.5, L.stream_ptr()) == 0
        assert rel(out_ws, out_1) < 2e-6
        out_h = torch.empty(B, N, C, device="cuda")             # the fp16-split kernel through the same key split
        assert lib.evc_attention_f16x3_f32(*args, L.fptr(out_h), C, B, heads, N, D, D ** -0.5,
                                           L.ptr(qkv_bounds(L, qkv, C)), L.ptr(ws), L.stream_ptr()) == 0
        assert rel(out_h, out_1) < 5e-6
        if B * N <= 2048:
            qh, kh, vh = [t.double().reshape(B, N, heads, D).permute(0, 2, 1, 3) for t in (q, k, v)]
            w = torch.softmax(torch.einsum("bhqd,bhkd->bhqk", qh, kh) * (D ** -0.5), dim=-1)
            ref = torch.einsum("bhqk,bhkd->bhqd", w, vh).permute(0, 2, 1, 3).reshape(B, N, C).float()
            assert rel(out_ws, ref) < 2e-5
    # a workspace is required when the plan splits
    assert lib.evc_attention_ws_f32(*args, L.fptr(out_ws), C, B, heads, N, D, D ** -0.5, None, L.stream_ptr()) == -1


@pytest.mark.parametrize("B,heads,N,D", [(9, 2, 1024, 192), (2, 3, 576, 192), (1, 2, 1000, 192), (2, 1, 512, 128), (3, 2, 640, 64),
                                         (2, 4, 520, 32), (2, 3, 256, 192)])
def test_attention_kv_planes_equal_the_in_kernel_conversion(L, B, heads, N, D):
    """evc_attention_f16x3_f32 with K / V converted ONCE per launch into tile images staged by LDS-DMA (option "kv_planes",
    the default) against the same kernel converting every tile in every query block: the same conversion code on the same
    numbers, so the outputs are EQUAL bit for bit -- incl. ragged last key tiles (N = 1000, 520), key parts + merge, every head
    width the fp16 kernel covers, and a shape below the 512 keys from which the pre-pass is used -- and both match fp64."""
    C = heads * D
    q = rnd(47, B, N, C)
    k = rnd(48, B, N, C) * torch.linspace(0.3, 3.0, N)[None, :, None]
    v = rnd(49, B, N, C)
    qkv = torch.cat([q, k, v], 2).cuda()
    bounds = qkv_bounds(L, qkv, C)
    assert L.hip_lib().evc_attention_workspace_bytes(B, heads, N, D) >= B * heads * ((N + 31) // 32) * (64 * (2 * D + 16) + 160 * D)
    try:
        L.attention_set_option("kv_planes", 0)
        base = L.attention(qkv, C, heads, bounds=bounds)
    finally:
        L.attention_set_option("kv_planes", 1)
    out = L.attention(qkv, C, heads, bounds=bounds)
    assert torch.equal(out, base)
    assert torch.equal(L.attention(qkv, C, heads, bounds=bounds), out)            # and deterministic
    if B * N <= 2048:
        qh, kh, vh = [t.double().reshape(B, N, heads, D).permute(0, 2, 1, 3) for t in (q, k, v)]
        w = torch.softmax(torch.einsum("bhqd,bhkd->bhqk", qh, kh) * (D ** -0.5), dim=-1)
        ref = torch.einsum("bhqk,bhkd->bhqd", w, vh).permute(0, 2, 1, 3).reshape(B, N, C).float()
        assert rel(out, ref) < 2e-5
    with pytest.raises(L.EvcKernelError):
        L.attention_set_option("no_such_option", 1)


@pytest.mark.parametrize("inverse,simplified", [(False, False), (True, False), (False, True), (True, True)])
def test_gdn_against_formula(L, inverse, simplified):
    """GDN / IGDN / GDN1 (ELICUtilis/layers/gdn.py:62-77, 95-106) against the formula in fp64, with compressai's
    NonNegativeParametrizer applied to raw parameters (some below the lower bound, to exercise the clamp)."""
    B, H, W, C = 2, 12, 10, 192
    x = rnd(90, B, C, H, W)
    beta_raw = (1.0 + 0.3 * rnd(91, C)).abs()
    beta_raw[:5] = 1e-5                                    # below sqrt(beta_min + pedestal): clamped
    gamma_raw = (0.1 * torch.eye(C) + 0.02 * rnd(92, C, C)).abs().sqrt()
    gamma_raw[0, :7] = 0.0                                 # below the bound: re-parametrises to exactly 0
    ped = (2.0 ** -18) ** 2
    beta = torch.clamp(beta_raw.double(), min=(1e-6 + ped) ** 0.5) ** 2 - ped
    gamma = torch.clamp(gamma_raw.double(), min=ped ** 0.5) ** 2 - ped
    xin = x.double().abs() if simplified else x.double() ** 2
    norm = F.conv2d(xin, gamma.reshape(C, C, 1, 1), beta)
    if simplified:
        ref = x.double() * (norm if inverse else 1.0 / norm)
    else:
        ref = x.double() * (torch.sqrt(norm) if inverse else torch.rsqrt(norm))
    op = L.GDN(beta_raw, gamma_raw, inverse=inverse, simplified=simplified)
    out = op(nhwc(x).cuda())
    assert rel(nchw(out), ref.float()) < 2e-6


@pytest.mark.parametrize("B,H,W,Ci,Co", [(2, 8, 8, 320, 192), (1, 16, 16, 192, 192), (3, 64, 64, 192, 3), (2, 2, 2, 192, 320)])
def test_deconv5x5s2_polyphase_vs_conv_transpose(L, B, H, W, Ci, Co):
    """evc_deconv5x5s2_f32 = compressai deconv(): ConvTranspose2d(k 5, stride 2, padding 2, output_padding 1), as one 3x3
    convolution with 4*Co phase-major outputs + depth-to-space (the g_s / h_s shapes of ELIC, incl. the 192 -> 3 output
    layer and the 2x2 hyper-latent)."""
    x = rnd(100, B, Ci, H, W)
    w = rnd(101, Ci, Co, 5, 5) / np.sqrt(Ci * 25 / 4)
    b = rnd(102, Co)
    ref = F.relu(F.conv_transpose2d(x, w, b, stride=2, padding=2, output_padding=1))
    op = L.Deconv5x5s2(w, b)
    out = op(nhwc(x).cuda(), act_out=L.ACT_RELU)
    assert out.shape == (B, 2 * H, 2 * W, Co)
    assert rel(nchw(out), ref) < 1e-5


@pytest.mark.parametrize("B,Ho,Wo,Ci,Co,ld", [(2, 32, 32, 3, 192, 16), (1, 16, 16, 192, 192, 192), (2, 4, 4, 192, 320, 192),
                                               (3, 1, 1, 192, 192, 192)])
def test_conv5x5s2_polyphase_vs_strided_conv(L, B, Ho, Wo, Ci, Co, ld):
    """evc_conv5x5s2_f32 = compressai conv(): Conv2d(k 5, stride 2, padding 2), as space-to-depth + one 3x3 convolution
    (the g_a / h_a shapes of ELIC, incl. the 3-channel image packed into a 16-wide NHWC row)."""
    x = rnd(103, B, Ci, 2 * Ho, 2 * Wo)
    w = rnd(104, Co, Ci, 5, 5) / np.sqrt(Ci * 25)
    b = rnd(105, Co)
    ref = F.conv2d(x, w, b, stride=2, padding=2)
    xin = torch.full((B, 2 * Ho, 2 * Wo, ld), 7.0)                 # padding channels hold junk: they must not be read
    xin[..., :Ci] = nhwc(x)
    out = L.Conv5x5s2(w, b)(xin.cuda(), channels=Ci)
    assert out.shape == (B, Ho, Wo, Co)
    assert rel(nchw(out), ref) < 1e-5


def test_layout_pack_and_unpack(L):
    B, H, W = 2, 6, 10
    x, c = rnd(50, B, 15, H, W).cuda(), rnd(51, B, 6, H, W).cuda()
    p = L.pack_nchw_to_nhwc(x, c, 32)
    ref = torch.zeros(B, H, W, 32)
    ref[..., :15] = nhwc(x).cpu()
    ref[..., 15:21] = nhwc(c).cpu()
    assert torch.equal(p.cpu(), ref)
    assert torch.equal(L.nhwc_to_nchw(p, 15).cpu(), x.cpu())
    assert torch.equal(L.pack_nchw_to_nhwc(x, None, 16).cpu()[..., :15], nhwc(x).cpu())


def test_sampler_step_kernels(L):
    n = (2, 15, 8, 8)
    x, e, z = rnd(60, *n).cuda(), rnd(61, *n).cuda(), rnd(62, *n).cuda()
    k1, k2, c1, c2, sg = 1.3, 0.7, 0.4, 0.55, 0.2
    x0 = (k1 * (x - k2 * e)).clip(-1, 1)
    ref = c1 * x0 + c2 * x + sg * z
    y = x.clone(); L.ddpm_step(y, e, z, k1, k2, c1, c2, sg, True)
    assert rel(y, ref) < 1e-6
    y = x.clone(); L.ddpm_step(y, e, None, k1, k2, c1, c2, sg, True)
    assert rel(y, c1 * x0 + c2 * x) < 1e-6
    y = x.clone(); L.ddim_step(y, e, k1, k2, c1, c2, False)
    assert rel(y, c1 * (k1 * (x - k2 * e)) + c2 * e) < 1e-6
    assert rel(L.axpy(x, e, -0.3), x - 0.3 * e) < 1e-6
    assert rel(L.pndm_transfer(x, e, 0.1, 0.8, 1.1, True), (x + 0.1 * (0.8 * x - 1.1 * e)).clip(-1, 1)) < 1e-6
    assert rel(L.lincomb4([x, e, z, x], [55 / 24, -59 / 24, 37 / 24, -9 / 24]),
               (55 * x - 59 * e + 37 * z - 9 * x) / 24) < 1e-6
    assert rel(L.lincomb4([x, e], [0.5, 2.0]), 0.5 * x + 2 * e) < 1e-6
    assert rel(L.scale_clamp(x, 0.5, 0.5, (0.0, 1.0)), ((x + 1) / 2).clamp(0, 1)) < 1e-6
    assert rel(L.gate_residual(x, e, z), x * torch.sigmoid(e) + z) < 1e-6


def test_elic_checkerboard_gather_scatter(L):
    B, H, W, C, ld = 2, 8, 8, 16, 40
    ms = torch.zeros(B, H, W, ld)
    means = rnd(70, B, H, W, C)
    scales = rnd(71, B, H, W, C).abs() * 3
    ms[..., 4:4 + C], ms[..., 24:24 + C] = means, scales
    table = torch.exp(torch.linspace(np.log(0.11), np.log(256), 64))
    for parity in (0, 1):
        idx, mu = L.elic_gather_params(ms.cuda(), 4, 24, C, parity, table.cuda())
        m_nchw, s_nchw = means.permute(0, 3, 1, 2), scales.permute(0, 3, 1, 2)
        m_enc, s_enc = torch.zeros(B, C, H, W // 2), torch.zeros(B, C, H, W // 2)
        o0, o1 = (0, 1) if parity == 0 else (1, 0)
        m_enc[:, :, 0::2], m_enc[:, :, 1::2] = m_nchw[:, :, 0::2, o0::2], m_nchw[:, :, 1::2, o1::2]
        s_enc[:, :, 0::2], s_enc[:, :, 1::2] = s_nchw[:, :, 0::2, o0::2], s_nchw[:, :, 1::2, o1::2]
        ref_idx = torch.full(s_enc.shape, 63, dtype=torch.int32)
        sc = s_enc.clamp(min=0.11)
        for s in table[:-1]:
            ref_idx -= (sc <= s).int()
        assert torch.equal(idx.cpu(), ref_idx)
        assert torch.equal(mu.cpu(), m_enc)
        sym = torch.randint(-5, 6, (B, C, H, W // 2), dtype=torch.int32)
        y = torch.zeros(B, H, W, 48, device="cuda")
        L.elic_scatter_symbols(sym.cuda(), mu, y, 32, parity)
        full = torch.zeros(B, C, H, W)
        q = sym.float() + m_enc
        full[:, :, 0::2, o0::2], full[:, :, 1::2, o1::2] = q[:, :, 0::2], q[:, :, 1::2]
        assert torch.equal(y.cpu()[..., 32:48].permute(0, 3, 1, 2), full)
        assert float(y.cpu()[..., :32].abs().max()) == 0.0


def test_clock_probe_reports_a_plausible_shader_clock():
    """``evc_clock_probe`` (bench.py's held-clock measurement): an idle wave counts shader-clock and 100 MHz reference ticks
    until the work enqueued behind it is done (stream-ordered stop word) or its time limit; the ratio is a clock the chip can
    run at."""
    import evc_amd  # noqa: F401
    from evc_amd import lib as L
    pr = L.ClockProbe(2_000_000)                 # at most 2 s ...
    x = torch.randn(1 << 22, device="cuda")
    for _ in range(200):
        x = L.scale_clamp(x, 1.0001, 0.0)        # ... but it ends with this work
    pr.stop()
    torch.cuda.synchronize()
    ghz = pr.ghz()
    assert 1e-4 < pr.seconds() < 0.5, pr.seconds()
    assert 0.1 < ghz < 2.6, ghz                  # MI355X: idle ~0.1-0.2 GHz, peak 2.4 GHz
    lim = L.ClockProbe(20000)                    # never stopped: runs to its 20 ms limit
    assert 0.019 <= (lim.ghz() and lim.seconds()) <= 0.04
    p = L.gpu_power_w()
    assert p is None or 50 < p < 1500
    assert L.hip_lib().evc_clock_probe(None, 10, None, None) != 0
    assert L.hip_lib().evc_clock_probe(pr.out.data_ptr(), 0, None, None) != 0


def test_im2col_and_maxpool_against_torch():
    """The two non-convolution layers of the LPIPS backbone: stride-4 11x11 patches (ScalingLayer fused, zero padding after
    the scaling, zero fill to the padded row length) vs ``F.unfold``; MaxPool2d(3, 2) vs ``F.max_pool2d``, odd sizes included."""
    import torch.nn.functional as F
    import evc_amd  # noqa: F401
    from evc_amd import lib as L
    x = rnd(11, 3, 3, 37, 45)
    shift, scale = torch.tensor([-0.03, -0.088, -0.188]), torch.tensor([0.458, 0.448, 0.45])
    got = L.im2col_nchw(x.cuda(), 11, 11, 4, 2, 368, shift.cuda(), scale.cuda()).cpu()
    xs = (x - shift.view(1, 3, 1, 1)) / scale.view(1, 3, 1, 1)
    ref = F.unfold(xs, kernel_size=11, stride=4, padding=2)                      # (N, 363, L), (c, ky, kx) order
    Ho, Wo = (37 + 4 - 11) // 4 + 1, (45 + 4 - 11) // 4 + 1
    assert got.shape == (3, Ho, Wo, 368)
    np.testing.assert_allclose(got[..., :363].reshape(3, Ho * Wo, 363).numpy(), ref.transpose(1, 2).numpy(), rtol=1e-6, atol=1e-6)
    assert float(got[..., 363:].abs().max()) == 0.0
    plain = L.im2col_nchw(x.cuda(), 3, 5, 2, 1, 48).cpu()                       # no scaling layer, rectangular kernel
    ref = F.unfold(x, kernel_size=(3, 5), stride=2, padding=1)
    np.testing.assert_array_equal(plain[..., :45].reshape(3, -1, 45).numpy(), ref.transpose(1, 2).numpy())
    for H, W in ((31, 31), (15, 15), (8, 13)):
        f = rnd(12, 2, 64, H, W)
        got = L.maxpool3s2_nhwc(f.permute(0, 2, 3, 1).contiguous().cuda()).cpu().permute(0, 3, 1, 2)
        np.testing.assert_array_equal(got.numpy(), F.max_pool2d(f, 3, 2).numpy())


def test_lpips_alexnet_against_the_oracle():
    """LPIPS(net='alex') (city_sender.py:302, :376-406) on the HIP kernels vs the CPU restatement of the algorithm the
    reference vendors (models/networks_basic.py:62-93, oracle/lpips.py) in float64, under the reference's own trained linear
    layers (weights/v0.1/alex.pth -> tests/golden/lpips_alex_lin.npz) and seeded stand-ins for the AlexNet convolutions
    (torchvision's checkpoint is not in the reference tree: backbone weights unpinned): per-pair distances, the single-frame
    call form of the reference, d(x, x) = 0, the key spellings of a saved LPIPS module."""
    import evc_amd  # noqa: F401
    from evc_amd.lpips import LpipsAlex
    from oracle import lpips as OL
    lin = golden("lpips_alex_lin")
    assert all(float(lin[f"lin{i}"].min()) >= 0.0 for i in range(5))          # the trained layers are non-negative
    sd = OL.seeded_state_dict(5, lin)
    net = LpipsAlex(sd)
    a = torch.rand(5, 3, 128, 128, generator=torch.Generator().manual_seed(1))
    b = (a + 0.1 * rnd(2, 5, 3, 128, 128)).clamp(0, 1)
    want = OL.distance({k: v.double() for k, v in sd.items()}, a.double(), b.double()).numpy()
    got = net(a.cuda(), b.cuda()).cpu().numpy()
    assert got.shape == (5,) and np.all(want > 1e-5)
    np.testing.assert_allclose(got, want, rtol=2e-4)
    one = net(a[2].cuda(), b[2].cuda())                                        # decide_5to5_lpips passes (3, H, W) frames
    assert one.shape == (1,) and abs(float(one) - want[2]) <= 2e-4 * want[2]
    assert float(net(a.cuda(), a.cuda()).abs().max()) == 0.0
    taps = net.features(a.cuda())
    ref_taps = OL.features(sd, a)
    for t, r in zip(taps, ref_taps):
        r = r.permute(0, 2, 3, 1).numpy()
        assert t.shape == r.shape and float(np.abs(t.cpu().numpy() - r).max()) <= 2e-5 * float(np.abs(r).max())
    renamed = {}
    for k, v in sd.items():
        if k.startswith("features."):
            idx = int(k.split(".")[1])
            renamed[{0: "net.slice1.0", 3: "net.slice2.3", 6: "net.slice3.6", 8: "net.slice4.8", 10: "net.slice5.10"}[idx] + "." + k.split(".")[2]] = v
        else:
            renamed["lins." + k[3:]] = v
    np.testing.assert_array_equal(LpipsAlex(renamed)(a.cuda(), b.cuda()).cpu().numpy(), got)
    with pytest.raises(KeyError):
        LpipsAlex({k: v for k, v in sd.items() if not k.startswith("lin3")})


def test_lpips_policy_metric_from_weight_files(tmp_path):
    """``--policy lpips --metric alexnet.pth,alex.pth``: the two weight files of the packages (here: seeded stand-ins saved in
    their layouts) load through the weights-only loader into the HIP metric, which then drives the reference's accept rule
    (distance <= threshold, decide_5to5_lpips)."""
    import evc_amd  # noqa: F401
    from evc_amd import policy as P
    from oracle import lpips as OL
    sd = OL.seeded_state_dict(6, golden("lpips_alex_lin"))
    torch.save({k: v for k, v in sd.items() if k.startswith("features.")}, tmp_path / "alexnet.pth")
    torch.save({k: v for k, v in sd.items() if k.startswith("lin")}, tmp_path / "alex.pth")
    m = P.load_metric("lpips", f"{tmp_path / 'alexnet.pth'},{tmp_path / 'alex.pth'}", "cuda")
    gt = torch.rand(4, 3, 128, 128, generator=torch.Generator().manual_seed(3))
    pred = gt.clone()
    pred[1:] += 0.05 * rnd(4, 3, 3, 128, 128)
    pred[3] = torch.rand(3, 128, 128, generator=torch.Generator().manual_seed(4))
    v = m.values(pred.cuda(), gt.cuda())
    want = OL.distance(sd, pred, gt).numpy()
    np.testing.assert_allclose(v, want, rtol=1e-3, atol=1e-7)
    assert v[0] == 0.0 and v[3] > v[1] > 0
    thr = float((v[2] + v[3]) / 2)
    assert [bool(m.accept(x, thr)) for x in v] == [True, True, True, False]
    torch.save(sd, tmp_path / "both.pth")                                       # one file holding everything
    np.testing.assert_array_equal(P.load_metric("lpips", str(tmp_path / "both.pth"), "cuda").values(pred.cuda(), gt.cuda()), v)


# ---- frame-axis kernels of the pseudo-3-D score network (csrc/frames.hip) -------------------------------------------------

def _video(seed, B, N, H, W, C):
    """(B*N, H, W, C) with a sample's frames adjacent + the same data as the reference's per-pixel (B*H*W, C, N)."""
    v = rnd(seed, B, N, H, W, C)
    return v.reshape(B * N, H, W, C).cuda(), v.permute(0, 2, 3, 4, 1).reshape(B * H * W, C, N)


@pytest.mark.parametrize("B,N,H,W,C,G", [(2, 5, 4, 4, 32, 8), (1, 3, 8, 8, 64, 16), (3, 7, 2, 2, 576, 32), (1, 1, 4, 4, 16, 4)])
def test_frame_group_norm_against_torch(L, B, N, H, W, C, G):
    """GroupNorm of AttnBlockpp1d (layers3d.py:89-90,107): per pixel over (C / G channels x N frames)."""
    x, ref_in = _video(301, B, N, H, W, C)
    gamma, beta = (1 + 0.1 * rnd(302, C)), 0.1 * rnd(303, C)
    ref = F.group_norm(ref_in.double(), G, gamma.double(), beta.double(), 1e-6)          # (B*H*W, C, N)
    y = L.frame_group_norm(x, N, gamma.cuda(), beta.cuda(), G, 1e-6)
    got = y.cpu().reshape(B, N, H, W, C).permute(0, 2, 3, 4, 1).reshape(B * H * W, C, N)
    assert rel(got, ref.float()) < 2e-6


@pytest.mark.parametrize("B,N,H,W,C,heads", [(2, 5, 4, 4, 64, 2), (1, 3, 8, 8, 32, 1), (2, 7, 2, 2, 384, 2), (1, 8, 2, 2, 96, 3)])
def test_frame_attention_against_torch(L, B, N, H, W, C, heads):
    """AttnBlockpp1d's attention (layers3d.py:112-118): softmax over the frames of one pixel, per head."""
    qkv, ref_in = _video(311, B, N, H, W, 3 * C)                                         # ref_in: (B*H*W, 3C, N)
    D = C // heads
    q, k, v = (ref_in[:, j * C:(j + 1) * C].double().reshape(-1, D, N) for j in range(3))
    w = torch.softmax(torch.einsum("bct,bci->bti", q, k) * (int(D) ** (-0.5)), dim=-1)
    ref = torch.einsum("bti,bci->bct", w, v).reshape(B * H * W, C, N)
    o = L.frame_attention(qkv, N, C, heads)
    got = o.cpu().reshape(B, N, H, W, C).permute(0, 2, 3, 4, 1).reshape(B * H * W, C, N)
    assert rel(got, ref.float()) < 2e-6


@pytest.mark.parametrize("B,N,M,H,W,C", [(2, 5, 3, 4, 4, 32), (1, 7, 5, 8, 8, 16), (3, 2, 2, 2, 2, 64), (1, 8, 1, 4, 4, 16)])
def test_frame_mix_against_torch(L, B, N, M, H, W, C):
    """The frame converters (ncsnpp_more.py:328-335,344-351): a 1x1 convolution over the frame axis."""
    x, _ = _video(321, B, N, H, W, C)
    w, b = rnd(322, M, N) / np.sqrt(N), 0.1 * rnd(323, M)
    ref = torch.einsum("bnhwc,mn->bmhwc", x.cpu().double().reshape(B, N, H, W, C), w.double()) + b.double()[None, :, None, None, None]
    y = L.frame_mix(x, N, w.cuda(), b.cuda())
    assert y.shape == (B * M, H, W, C)
    assert rel(y.cpu().reshape(B, M, H, W, C), ref.float()) < 2e-6


def test_time_convolution_as_a_kx1_filter_over_the_frame_axis(L, arith):
    """PseudoConv3d.time_conv (layers3d.py:274,294-297): Conv1d over the frames of each pixel = the 2-D convolution with a
    3 x 1 filter over B "images" of N rows x H*W columns (zero padding along the frames only), SiLU on load, residual."""
    B, N, H, W, C, Co = 2, 5, 4, 4, 32, 48
    x, ref_in = _video(331, B, N, H, W, C)                                               # ref_in: (B*H*W, C, N)
    res, res_ref = _video(332, B, N, H, W, Co)
    w, b = rnd(333, Co, C, 3) / np.sqrt(3 * C), 0.1 * rnd(334, Co)
    ref = (F.conv1d(F.silu(ref_in.double()), w.double(), b.double(), padding=1) + res_ref.double()) * 0.70710678
    wp = L.conv_pack_weights(w[:, :, :, None].contiguous().cuda(), arith)
    y = L.conv2d_nhwc(x.view(B, N, H * W, C), wp, Co, 3, 1, bias=b.cuda(), act_in=L.ACT_SILU,
                      res=res.view(B, N, H * W, Co), out_scale=0.70710678)
    got = y.cpu().reshape(B, N, H, W, Co).permute(0, 2, 3, 4, 1).reshape(B * H * W, Co, N)
    assert rel(got, ref.float()) < 1e-5


def test_conv3d_as_frame_taps_plus_one_3x3_convolution(L, arith):
    """nn.Conv3d 3x3x3 over (N, H, W) (MyConv3d, layers3d.py:225-243) = evc_frame_taps_f32 (frames n - 1 | n | n + 1 side by
    side along the channels, zeros beyond a sample's frames) + ONE 3x3 convolution with 3C input channels and the weight read
    as (Co, kt*Ci + ci, kh, kw), against torch.nn.functional.conv3d."""
    B, N, H, W, C, Co = 2, 4, 6, 6, 16, 32
    v = rnd(341, B, N, H, W, C)
    x = v.reshape(B * N, H, W, C).cuda()
    w, b = rnd(342, Co, C, 3, 3, 3) / np.sqrt(27 * C), 0.1 * rnd(343, Co)
    ref = F.conv3d(v.permute(0, 4, 1, 2, 3).double(), w.double(), b.double(), padding=1)        # (B, Co, N, H, W)
    taps = L.frame_taps(x, N)
    assert taps.shape == (B * N, H, W, 3 * C)
    t5 = taps.cpu().reshape(B, N, H, W, 3, C)
    assert torch.equal(t5[:, :, :, :, 1], v) and torch.equal(t5[:, 1:, :, :, 0], v[:, :-1]) and torch.equal(t5[:, :-1, :, :, 2], v[:, 1:])
    assert float(t5[:, 0, :, :, 0].abs().max()) == 0 and float(t5[:, -1, :, :, 2].abs().max()) == 0
    w2 = w.permute(0, 2, 1, 3, 4).reshape(Co, 3 * C, 3, 3).contiguous().cuda()
    y = L.conv2d_nhwc(taps, L.conv_pack_weights(w2, arith), Co, 3, 3, bias=b.cuda())
    assert rel(y.cpu().reshape(B, N, H, W, Co).permute(0, 4, 1, 2, 3), ref.float()) < 1e-5
