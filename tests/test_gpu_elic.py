"""GPU parity tests for the ELIC key-frame codec (HIP) against the CPU oracle restatement.

Integer symbols must be exact ACROSS IMPLEMENTATIONS: the entropy-parameter networks run under EVC_ARITH_F32 (one
fixed-order chain of fmaf per output), which the oracle restates bit for bit (oracle/exact_conv.c), so a stream written
by either implementation decodes to the same symbols, and the same bytes come out of both encoders given the same
latents (``test_cross_implementation_*``).  The remaining float tensors (g_a, g_s) are compared at fp32 tolerance."""
import numpy as np
import pytest
import torch

from conftest import rnd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    import evc_amd  # noqa: F401
    from evc_amd import synthetic
    from evc_amd.elic import ElicModel
    sd = synthetic.elic_state_dict(21)
    frames = synthetic.make_clips(1, seed=3, frames=3, size=128)[0].astype(np.float32) / 255
    return sd, ElicModel(sd), torch.from_numpy(frames)


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def test_encode_decode_symbols_exact_and_batch_independent(setup):
    sd, model, x = setup
    enc = model.compress(x, return_latents=True)
    assert enc["shape"] == (2, 2) and len(enc["strings"][0]) == 5 and len(enc["strings"][1]) == 3
    dec = model.decompress(enc["strings"], enc["shape"], return_latents=True)
    assert torch.equal(dec["y_hat"], enc["y_hat"])                      # decoder == encoder latents, bitwise
    assert float((enc["y_hat"] - enc["y"]).abs().max()) <= 0.5 + 1e-4  # rounding residual
    assert dec["x_hat"].shape == (3, 3, 128, 128)
    assert float(dec["x_hat"].min()) >= 0.0 and float(dec["x_hat"].max()) <= 1.0
    # decode frame 1 alone: same bytes -> bit-identical pixels (no dependence on batch composition)
    ys = [[[p[1]] for p in sl] for sl in enc["strings"][0]]
    one = model.decompress([ys, [enc["strings"][1][1]]], enc["shape"])
    assert torch.equal(one["x_hat"][0], dec["x_hat"][1])
    # encoding frame 1 alone gives the same bytes as inside the batch
    enc1 = model.compress(x[1:2])
    assert enc1["strings"][1][0] == enc["strings"][1][1]
    assert all(enc1["strings"][0][i][p][0] == enc["strings"][0][i][p][1] for i in range(5) for p in range(2))


def test_against_oracle_with_teacher_forced_symbols(setup):
    from evc_amd import lib
    from oracle import elic as OEL
    sd, model, x = setup
    x = x[:2]
    enc = model.compress(x, return_latents=True)
    dec = model.decompress(enc["strings"], enc["shape"], return_latents=True)
    # analysis transform (encoder side) against the oracle
    assert rel(enc["y"], OEL.g_a(sd, x)) < 2e-4

    class Forced:
        """z strings are really decoded (integer path); y symbols are the ones the HIP decoder produced."""
        def __init__(self):
            self.queue = [s[b] for s in dec["symbols"] for b in range(s.shape[0])]
            self.z_left = 2

        def decode_with_indexes(self, s, idx, cdf, size, off):
            if self.z_left > 0:
                self.z_left -= 1
                return lib.rans_decode(s, idx, cdf, size, off)
            return self.queue.pop(0).reshape(-1)
    ref = OEL.decompress(sd, enc["strings"], enc["shape"], coder=Forced(), return_latents=True)
    assert torch.equal(ref["z_hat"], dec["z_hat"].cpu())        # integer z path: exact
    assert rel(dec["y_hat"], ref["y_hat"]) < 1e-4               # symbols + predicted means
    assert rel(dec["x_hat"], ref["x_hat"]) < 5e-4               # synthesis transform g_s, fp32 tolerance


def test_inference_wrapper_pad_crop_and_bits(setup):
    from evc_amd.elic import count_bits, inference
    sd, model, x = setup
    img = x[0][:, :100, :120]
    x_hat, bits = inference(model, img, patch=64)
    assert x_hat.shape == (1, 3, 100, 120) and bits > 0 and bits % 32 == 0
    enc = model.compress(torch.nn.functional.pad(img[None], (0, 8, 0, 28)))
    assert bits == count_bits(enc["strings"])


def test_receiver_refuses_stream_coded_under_another_revision(setup):
    """The container records the arithmetic and revision the encoder's entropy-parameter networks ran with -- always
    EVC_ARITH_F32 (a CPU-reproducible fmaf chain), whatever EVC_CONV_ARITH selects for the other transforms; a receiver
    refuses a stream whose tag differs (CodecMismatch) instead of desynchronising its range decoder."""
    import evc_amd  # noqa: F401
    from evc_amd import container, lib
    from evc_amd.elic import ELIC_CODEC_REV
    sd, model, x = setup
    assert model.codec_tag() == (lib.ARITH_F32, ELIC_CODEC_REV)
    enc = model.compress(x[:2])
    d = np.zeros(30, dtype=np.int64)
    d[0] = 1
    for foreign in ((lib.ARITH_BF16X6, ELIC_CODEC_REV), (lib.ARITH_F32, ELIC_CODEC_REV - 1)):
        blob = container.pack(d, [enc["strings"]], enc["shape"], codec=foreign)
        with pytest.raises(container.CodecMismatch):
            container.unpack(blob, expect_codec=model.codec_tag())
    blob = container.pack(d, [enc["strings"]], enc["shape"], codec=model.codec_tag())
    d2, keys, shape = container.unpack(blob, expect_codec=model.codec_tag())      # the matching receiver decodes
    out = model.decompress(keys[0], shape)["x_hat"]
    assert torch.equal(out, model.decompress(enc["strings"], enc["shape"])["x_hat"])


def _native_coder():
    from evc_amd import lib

    class Coder:
        encode_with_indexes = staticmethod(lib.rans_encode)
        decode_with_indexes = staticmethod(lib.rans_decode)
    return Coder


@pytest.mark.parametrize("q", [0, 1, 2, 3, 4, 5])
def test_cross_implementation_symbols_and_bytes_without_teacher_forcing(q):
    """north_star: "bit-exact for the ELIC entropy-decoded integer symbols".  Per quality index q0..q5 (six seeded models, the
    reference's six checkpoints: city_sender.py:478-484), 12 synthetic 64x64 frames (72 in all; 64x64 = one ELIC patch,
    Inference.py:21-31) go through BOTH directions with NO teacher forcing:
      (a) streams written by the HIP encoder are decoded by the CPU oracle (oracle/elic.decompress, exact mode): every
          symbol of every (slice, pass, frame) equals the HIP decoder's, z_hat and y_hat are bitwise equal;
      (b) streams written by the oracle encoder are decoded by the HIP decoder: y_hat equals the oracle's bit for bit;
      (c) given the SAME latents y (the HIP analysis transform's), the oracle encoder produces the same bytes as the HIP
          encoder, string by string.
    Mismatch counts are reported per slice; all must be zero.  Reference: Network.py:336-532."""
    import evc_amd  # noqa: F401
    from evc_amd import synthetic
    from evc_amd.elic import GROUPS, ElicModel
    from oracle import elic as OEL
    torch.set_num_threads(8)
    sd = synthetic.elic_state_dict(q)
    model = ElicModel(sd)
    coder = _native_coder()
    n = 12
    frames = torch.from_numpy(synthetic.make_clips(1, seed=40 + q, frames=n, size=64)[0].astype(np.float32) / 255)
    enc = model.compress(frames, return_latents=True)
    dec = model.decompress(enc["strings"], enc["shape"], return_latents=True)
    mism = {f"slice{i}": 0 for i in range(len(GROUPS) - 1)}
    total = 0
    # (a) HIP stream -> oracle decoder
    ref = OEL.decompress(sd, enc["strings"], enc["shape"], coder=coder, return_latents=True, exact=True)
    k = 0
    for i in range(len(GROUPS) - 1):
        for parity in (0, 1):
            hip_sym = dec["symbols"][i * 2 + parity]                      # (B, g, H, W/2)
            for b in range(n):
                o = ref["symbols"][k].reshape(-1)
                k += 1
                h = hip_sym[b].reshape(-1)
                mism[f"slice{i}"] += int((o != h).sum())
                total += h.size
    assert torch.equal(ref["z_hat"], dec["z_hat"].cpu())
    print(f"q{q}: HIP-encoded streams decoded by the oracle: {total} symbols, mismatches per slice {mism}")
    assert sum(mism.values()) == 0, mism
    assert torch.equal(ref["y_hat"], dec["y_hat"].cpu())                  # symbols + bit-identical predicted means
    # (b) oracle stream -> HIP decoder
    oenc = OEL.compress(sd, frames, coder=coder, exact=True)
    hdec = model.decompress(oenc["strings"], tuple(oenc["shape"]), return_latents=True)
    bad = int((hdec["y_hat"].cpu() != oenc["y_hat"]).sum())
    print(f"q{q}: oracle-encoded streams decoded by HIP: y_hat elements differing {bad} of {oenc['y_hat'].numel()}")
    assert bad == 0
    # (c) same latents -> same bytes: feed the HIP encoder's y and z-path to the oracle's entropy coder
    ore = _oracle_encode_from_latents(OEL, sd, enc["y"].cpu(), enc["strings"][1], enc["shape"], coder)
    for i in range(len(GROUPS) - 1):
        for parity in (0, 1):
            for b in range(n):
                assert ore[i][parity][b] == enc["strings"][0][i][parity][b], (q, i, parity, b)


def _oracle_encode_from_latents(OEL, p, y, z_strings, shape, coder):
    """The oracle's slice / checkerboard encoder loop (oracle/elic.compress, Network.py:352-441) on GIVEN latents y and a given
    z stream, exact mode: what the bytes must be for these latents."""
    from oracle import entropy as E
    B, C, H, W = y.shape
    cdf, cdf_len, off = OEL._tables(p, "entropy_bottleneck")
    zc = cdf.shape[0]
    med = p["entropy_bottleneck.quantiles"][:, 0, 1].reshape(-1, 1, 1)
    z_idx = np.broadcast_to(np.arange(zc, dtype=np.int32)[:, None, None], (zc, shape[0], shape[1])).reshape(-1)
    z_hat = torch.zeros(B, zc, shape[0], shape[1])
    for b in range(B):
        d = np.asarray(coder.decode_with_indexes(z_strings[b], z_idx.tolist(), cdf, cdf_len, off), dtype=np.float32)
        z_hat[b] = torch.from_numpy(d.reshape(zc, shape[0], shape[1])) + med
    lm, ls = OEL.h_s(p, z_hat, True).chunk(2, 1)
    gcdf, gcdf_len, goff = OEL._tables(p, "gaussian_conditional")
    table = p["gaussian_conditional.scale_table"].numpy()
    y_slices = torch.split(y, OEL.GROUPS[1:], 1)
    out, y_hat_slices = [], []
    for i, ys in enumerate(y_slices):
        g = OEL.GROUPS[i + 1]
        support = OEL._support(p, i, y_hat_slices, lm, ls, True)
        y_hat_i = torch.zeros_like(ys)
        ctx = torch.zeros(B, 2 * g, H, W)
        strs_i = []
        for parity in (0, 1):
            if parity == 1:
                ctx = OEL._ctx(p, i, y_hat_i, True)
            means, scales = OEL._pa(p, i, torch.cat([ctx, support], dim=1), True).chunk(2, 1)
            m_enc, s_enc, y_enc = OEL._pack(means, parity), OEL._pack(scales, parity), OEL._pack(ys, parity)
            idx = E.build_indexes(s_enc.numpy(), table)
            qv = torch.zeros_like(m_enc)
            strs = []
            for b in range(B):
                sym = E.quantize_symbols(y_enc[b].numpy(), m_enc[b].numpy()).reshape(-1)
                strs.append(coder.encode_with_indexes(sym.tolist(), idx[b].reshape(-1).tolist(), gcdf, gcdf_len, goff))
                qv[b] = torch.from_numpy(sym.astype(np.float32).reshape(m_enc[b].shape)) + m_enc[b]
            strs_i.append(strs)
            y_hat_i = y_hat_i + OEL._unpack(qv, parity, W)
        out.append(strs_i)
        y_hat_slices.append(y_hat_i)
    return out


@pytest.mark.parametrize("name,Ci,Co,K", [("h_s.4", 320, 640, 3), ("cc_transforms", 224, 128, 5), ("ParamAggregation", 1024, 640, 1)])
def test_entropy_parameter_convs_are_bitwise_batch_invariant(name, Ci, Co, K):
    """Encoder and decoder may run the entropy-parameter convolutions at different batch sizes (the receiver decodes
    key-frame runs at 2B): tile height, row-reuse kernel and 256-pixel tiles are chosen from the grid size, so the
    per-sample outputs must be BITWISE equal across every such switch, for both arithmetics ELIC can run with."""
    import evc_amd  # noqa: F401
    from evc_amd import lib as L
    w = (rnd(300, Co, Ci, K, K) / np.sqrt(Ci * K * K)).cuda()
    b = rnd(301, Co).cuda()
    for arith in (L.ARITH_BF16X6, L.ARITH_F32):
        wp = L.conv_pack_weights(w, arith)
        for (H, W) in ((8, 8), (32, 32)):
            x1 = rnd(302, 1, H, W, Ci).cuda()
            ref = L.conv2d_nhwc(x1, wp, Co, K, K, bias=b, act_out=L.ACT_RELU, splits=1)
            for B in (2, 3, 9, 18, 64):       # 8x8: M = 128 .. 4096 pixels; 32x32: 2048 .. 65536 (256-pixel tiles)
                xb = torch.cat([rnd(310 + i, 1, H, W, Ci) for i in range(B - 1)] + [x1.cpu()], 0).cuda()
                out = L.conv2d_nhwc(xb, wp, Co, K, K, bias=b, act_out=L.ACT_RELU, splits=1)
                assert torch.equal(out[-1], ref[0]), (name, arith, H, B)
