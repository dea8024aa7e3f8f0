"""GPU parity tests, network and sampler level: the HIP score network and sampling loops against the
golden vectors produced by the reference itself (tests/golden/make_goldens.py) and against the CPU oracle
on the same seeded weights and inputs."""
import numpy as np
import pytest
import torch

from conftest import gamma_feed, golden, rnd

pytestmark = pytest.mark.gpu


def make_config(ngf, head, image_size):
    import evc_amd  # noqa: F401
    from evc_amd.config import default_config
    return default_config(ngf, head, image_size)


def build(ngf, head, image_size, seed):
    import evc_amd  # noqa: F401
    from evc_amd.scorenet import ScoreNet
    from oracle.scorenet import Dims, seeded_params
    d = Dims(ngf=ngf, n_head_channels=head, image_size=image_size)
    p = seeded_params(d, seed)
    return ScoreNet(make_config(ngf, head, image_size), p), d, p


def rel(a, b):
    a = a.detach().float().cpu().numpy() if torch.is_tensor(a) else a
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def test_forward_ngf64_against_reference_golden():
    net, d, p = build(64, 64, 32, 21)
    x, cond = rnd(22, 2, 15, 32, 32).cuda(), rnd(23, 2, 6, 32, 32).cuda()
    out = net(x, torch.tensor([430, 430]), cond=cond)
    assert out.shape == (2, 15, 32, 32)
    assert rel(out, golden("blocks_ngf64")["out"]) < 1e-4   # fp32, tolerance of SURVEY.md 8c


def test_every_module_output_of_the_hip_network_against_reference_taps():
    """GPU-side per-module parity: the output of every res-block, attention block and the two 3x3 convolutions of the HIP
    network (test hook ``ScoreNet.taps``) against the forward-hook taps of the reference's own modules
    (tests/golden/make_goldens.py::gen_blocks, ncsnpp_more.py:251-392), at the tolerances the CPU oracle is held to
    (tests/test_oracle_goldens.py:48-63: 2e-5) widened to 1e-4 for the fp32-equivalent split arithmetic."""
    from oracle.scorenet import program
    g = golden("blocks_ngf64")
    net, d, p = build(64, 64, 32, 21)
    x, cond = rnd(22, 2, 15, 32, 32).cuda(), rnd(23, 2, 6, 32, 32).cuda()
    net.taps = {}
    try:
        out = net(x, torch.tensor([430, 430]), cond=cond)
        taps = net.taps
    finally:
        net.taps = None
    assert rel(out, g["out"]) < 1e-4
    mods = program(d)
    checked, worst = 0, 0.0
    for idx, m in enumerate(mods):
        if m["kind"] not in ("res", "attn", "conv3"):
            continue
        assert idx in taps, (idx, m)
        t = taps[idx].float().cpu()
        flat = t.reshape(-1)
        stride = max(1, flat.numel() // 512)
        err = rel(flat[::stride][:512].numpy(), g[f"tap{idx}"])
        worst = max(worst, err)
        assert err < 1e-4, (idx, m, err)
        mean, std = g[f"tapstat{idx}"]
        assert abs(float(t.mean()) - float(mean)) < 1e-4 * float(np.abs(g[f"tap{idx}"]).max()), (idx, m)
        assert abs(float(t.std()) - float(std)) < 1e-4 * float(std), (idx, m)
        checked += 1
    print(f"{checked} module outputs checked, worst relative error {worst:.2e}")
    assert checked == sum(1 for m in mods if m["kind"] in ("res", "attn", "conv3")) == 47


def test_forward_ngf32_three_labels_incl_fractional_and_mixed_batch():
    g = golden("forward_ngf32")
    net, d, p = build(32, 32, 32, 31)
    x, cond = rnd(32, 2, 15, 32, 32).cuda(), rnd(33, 2, 6, 32, 32).cuda()
    for key, lab in (("out_t0", [0, 0]), ("out_t990", [990, 990]), ("out_tm05", [-0.5, -0.5])):
        assert rel(net(x, torch.tensor(lab), cond=cond), g[key]) < 1e-4, key
    mixed = net(x, torch.tensor([0, 990]), cond=cond)     # per-sample labels through the row table
    assert rel(mixed[0], g["out_t0"][0]) < 1e-4 and rel(mixed[1], g["out_t990"][1]) < 1e-4


@pytest.fixture(scope="module")
def full_net():
    """The full-size (262 M parameter) network with the golden generator's seed, built once for the module."""
    net, d, p = build(192, 192, 128, 1234)
    return net


@pytest.fixture(autouse=True)
def no_range_events():
    """Every network / trajectory test of this module must leave the fp16-split range-event word at zero: no tensor held
    a NaN / inf and every GroupNorm-ed operand was provably inside fp16's range (include/evc_hip.h EVC_RANGE_*)."""
    import evc_amd  # noqa: F401
    from evc_amd import lib as L
    L.range_events(reset=True)
    yield
    assert L.range_events() == 0


def test_forward_full_size_against_reference_golden(full_net):
    g = golden("forward_full")
    net = full_net
    x, cond = rnd(51, 1, 15, 128, 128).cuda(), rnd(52, 1, 6, 128, 128).cuda()
    o = net(x, torch.tensor([500]), cond=cond).cpu()
    assert rel(o.reshape(-1)[::60].numpy(), g["samples"]) < 2e-4
    assert rel(o[0, :, 0, :].numpy(), g["first_row"]) < 2e-4
    # batch invariance at full size: B=3 with identical samples gives identical outputs per sample
    o3 = net(x.repeat(3, 1, 1, 1), torch.tensor([500] * 3), cond=cond.repeat(3, 1, 1, 1)).cpu()
    assert rel(o3[2], o[0].numpy()) < 1e-5


def _b9_inputs():
    x = torch.cat([rnd(600 + i, 1, 15, 128, 128) for i in range(9)], 0).cuda()
    cond = torch.cat([rnd(700 + i, 1, 6, 128, 128) for i in range(9)], 0).cuda()
    return x, cond


def test_forward_full_size_b9_one_launch_against_reference_golden(full_net):
    """BASELINE configs[1]: the benchmark's own launch configuration -- nine different samples in ONE B=9 forward
    (tile heights / split-K factors are chosen per batch size) -- against the reference run on the same batch."""
    g = golden("forward_full_b9")
    x, cond = _b9_inputs()
    o = full_net(x, torch.tensor([500] * 9), cond=cond).cpu().reshape(9, -1)
    for i in range(9):
        assert rel(o[i, ::60].numpy(), g["samples"][i]) < 2e-4, i
    st = g["stats"]
    full = o.reshape(9, -1)
    assert np.allclose(full.mean(1).numpy(), st[:, 0], atol=2e-4 * float(st[:, 2].max()))
    assert np.allclose(full.std(1).numpy(), st[:, 1], rtol=2e-4)
    # configs[4] (F-PNDM) feeds fractional / negative labels, DDPM's denoise call the count label 99: same B=9 launch
    for key, lab in (("samples_tm05", -0.5), ("samples_t99", 99)):
        o = full_net(x, torch.tensor([lab] * 9), cond=cond).cpu().reshape(9, -1)
        for i in range(2):
            assert rel(o[i, ::60].numpy(), g[key][i]) < 2e-4, (key, i)


def test_forward_full_size_b32_cycled_against_reference_golden(full_net):
    """BASELINE configs[4] batches 32 clips per launch: 256-pixel 8-wave tiles and other split choices than B=9.
    32 inputs = the nine golden samples cycled; every output must match its sample's reference values."""
    g = golden("forward_full_b9")
    x, cond = _b9_inputs()
    idx = torch.arange(32) % 9
    o = full_net(x[idx.cuda()], torch.tensor([500] * 32), cond=cond[idx.cuda()]).cpu().reshape(32, -1)
    for j in range(32):
        assert rel(o[j, ::60].numpy(), g["samples"][j % 9]) < 2e-4, j


@pytest.mark.parametrize("B", [5, 6, 8])
def test_forward_full_size_per_rank_batches_against_reference_golden(full_net, B):
    """BASELINE configs[2] shards 46 clips as 6,6,6,6,6,6,5,5: a rank launches the network at B = 6 or 5 (and the CLI's
    default --batch is 8).  Tile height, split-K factor and K-split tail are chosen per batch size, so each of these
    launch plans is checked on its own: the first B samples of the B=9 golden batch in ONE launch of size B (the
    reference is batch-independent, models/better/ncsnpp_more.py:251-392, so the B=9 goldens pin every slice)."""
    g = golden("forward_full_b9")
    x, cond = _b9_inputs()
    o = full_net(x[:B].contiguous(), torch.tensor([500] * B), cond=cond[:B].contiguous()).cpu().reshape(B, -1)
    for i in range(B):
        assert rel(o[i, ::60].numpy(), g["samples"][i]) < 2e-4, (B, i)
    st = g["stats"][:B]
    assert np.allclose(o.mean(1).numpy(), st[:, 0], atol=2e-4 * float(st[:, 2].max()))
    assert np.allclose(o.std(1).numpy(), st[:, 1], rtol=2e-4)
    # the fractional F-PNDM label and DDPM's denoise label through the same launch plan
    for key, lab in (("samples_tm05", -0.5), ("samples_t99", 99)):
        o = full_net(x[:B].contiguous(), torch.tensor([lab] * B), cond=cond[:B].contiguous()).cpu().reshape(B, -1)
        for i in range(2):
            assert rel(o[i, ::60].numpy(), g[key][i]) < 2e-4, (B, key, i)


def test_every_conv_launch_of_the_full_size_forward_against_torch(full_net):
    """Walk every DISTINCT convolution launch the full-size forward makes at the benchmark's batch sizes (B = 9:
    configs[1]; B = 32: configs[4]; B = 5, 6: a rank of configs[2]; B = 8: the CLI's default batch) -- shape, source
    split, on-load transform, residual, arithmetic -- and run each through the C ABI exactly as the network does (same
    tile / split-K / kernel choice, which depend on B) against torch's fp32 arithmetic on the same device."""
    import torch.nn.functional as F
    from evc_amd import lib as L

    def conv_ref(x, w, bias, K):
        """torch fp32 convolution of an NHWC tensor in shift-and-matmul form (one fp32 GEMM per filter tap): the same numbers
        as F.conv2d up to summation order, without MIOpen's per-shape kernel search and JIT -- > 200 distinct shapes cost
        1.5 - 3 minutes of it on a fresh box, which is what made this test's duration vary from box to box."""
        Bx, Hx, Wx, _ = x.shape
        xp = F.pad(x, (0, 0, K // 2, K // 2, K // 2, K // 2))
        out = None
        for ty in range(K):
            for tx in range(K):
                t = xp[:, ty:ty + Hx, tx:tx + Wx, :] @ w[:, :, ty, tx].t()
                out = t if out is None else out + t
        return out if bias is None else out + bias
    seen = {}
    for B in (5, 6, 8, 9, 32):
        x = rnd(900, B, 15, 128, 128).cuda()
        c = rnd(901, B, 6, 128, 128).cuda()
        prof = []
        L.CONV_PROFILE = prof
        try:
            full_net.forward_label(x, 500, c)
        finally:
            L.CONV_PROFILE = None
        torch.cuda.synchronize()
        for r in prof:
            k = tuple(sorted(r["call"].items()))
            seen.setdefault(k, r["call"])
    assert len(seen) > 200, len(seen)                      # 5 batch sizes x ~50 distinct layer configurations
    worst = 0.0
    for n, call in enumerate(seen.values()):
        B, H, W, C0, C1, Co, K = (call[k] for k in ("B", "H", "W", "C0", "C1", "Co", "K"))
        C = C0 + C1
        g = torch.Generator(device="cuda").manual_seed(1000 + n)
        x0 = torch.randn(B, H, W, C0, device="cuda", generator=g)
        x1 = torch.randn(B, H, W, C1, device="cuda", generator=g) if C1 else None
        w = torch.randn(Co, C, K, K, device="cuda", generator=g) / np.sqrt(C * K * K)
        bias = torch.randn(Co, device="cuda", generator=g) if call["bias"] else None
        res = torch.randn(B, H, W, Co, device="cuda", generator=g) if call["res"] else None
        coef = None
        if call["coef"]:
            coef = (1 + 0.2 * torch.randn(B, C, device="cuda", generator=g), 0.3 * torch.randn(B, C, device="cuda", generator=g))
        bound = None
        if call["bound"]:
            bound = torch.zeros(1, dtype=torch.int32, device="cuda")
            L.gn_coeffs([L.chan_stats(x0)] + ([L.chan_stats(x1)] if C1 else []), H * W, 32 if C % 32 == 0 else 1, 1e-5,
                        bound=bound)
        wp = L.conv_pack_weights(w, call["arith"])
        out = torch.empty(B, H, W, call["ld_out"], device="cuda")
        x2 = x2t = w2 = None
        if call.get("x2"):           # a res-block's 1x1 skip convolution fused into its Conv_1 launch
            C2 = call["x2"]
            x2t = 3.0 * torch.randn(B, H, W, C2, device="cuda", generator=g)
            w2 = torch.randn(Co, C2, 1, 1, device="cuda", generator=g) / np.sqrt(C2)
            b2 = torch.zeros(1, dtype=torch.int32, device="cuda")
            L.gn_coeffs([L.chan_stats(x2t)], H * W, 32 if C2 % 32 == 0 else 1, 1e-5, bound=b2)
            x2 = (x2t, None, L.conv_pack_weights(w2, L.ARITH_F16X3), b2)
        got = L.conv2d_nhwc(x0, wp, Co, K, K, bias=bias, src1=x1, coef=coef, act_in=call["act_in"], res=res,
                            out_scale=call["out_scale"], out=out, in_bound=bound, want_stats=call["stats"], x2=x2)
        got = got[0] if call["stats"] else got
        xin = torch.cat([x0, x1], 3) if C1 else x0
        if coef is not None:
            xin = xin * coef[0][:, None, None, :] + coef[1][:, None, None, :]
        if call["act_in"] == L.ACT_SILU:
            xin = F.silu(xin)
        ref = conv_ref(xin, w, bias, K)
        if x2 is not None:
            ref = ref + conv_ref(x2t, w2, None, 1)
        if res is not None:
            ref = ref + res
        ref = ref * call["out_scale"]
        err = float((got[..., :Co] - ref).abs().max() / ref.abs().max())
        worst = max(worst, err)
        assert err < 3e-5, (call, err)
    print(f"{len(seen)} distinct conv launches, worst relative error {worst:.2e}")


def test_full_size_ddpm_trajectory_against_reference_golden(full_net):
    """5 DDPM steps + the denoise call (6 full-size forwards, B=2, injected noise) against the reference sampler."""
    import evc_amd  # noqa: F401
    from evc_amd import sampler
    g = golden("traj_full")
    x_T, cond = rnd(801, 2, 15, 128, 128).cuda(), rnd(802, 2, 6, 128, 128).cuda()
    noises = [rnd(810 + i, 2, 15, 128, 128) for i in range(5)]
    out = sampler.ddpm_sampler(x_T, full_net, cond=cond, subsample_steps=5, denoise=True, clip_before=True,
                               final_only=True, noise_fn=lambda i, x: noises[i])[0].cpu()
    assert rel(out.reshape(2, -1)[:, ::30].numpy(), g["samples"]) < 5e-4
    assert rel(out[:, :, 0, :].numpy(), g["first_row"]) < 5e-4


def test_full_size_fpndm_trajectory_against_reference_golden(full_net):
    """BASELINE configs[4]'s sampler at full size: F-PNDM, 10 subsampled steps = 3 Runge-Kutta warm-up iterations (4
    forwards each, fractional midpoint labels) + 7 Adams-Bashforth ones = 19 full-size forwards, B=2, no noise, against
    the reference sampler (models/__init__.py:39-100, models/pndm.py:3-52) run on the same weights and inputs."""
    import evc_amd  # noqa: F401
    from evc_amd import sampler
    g = golden("traj_fpndm_full")
    x_T, cond = rnd(821, 2, 15, 128, 128).cuda(), rnd(822, 2, 6, 128, 128).cuda()
    out = sampler.FPNDM_sampler(x_T, full_net, cond=cond, subsample_steps=10, final_only=True, clip_before=True)[0].cpu()
    assert rel(out.reshape(2, -1)[:, ::30].numpy(), g["samples"]) < 5e-4
    assert rel(out[:, :, 0, :].numpy(), g["first_row"]) < 5e-4
    st = g["stats"]
    assert abs(float(out.std()) - float(st[1])) < 5e-4 * float(st[2])


def test_full_size_ddim_trajectory_against_reference_golden(full_net):
    """5 deterministic DDIM steps + the denoise call (6 full-size forwards, B=2) against the reference sampler
    (models/__init__.py:103-204) run on the same weights and inputs (tests/golden/make_goldens.py::gen_traj_ddim_full)."""
    import evc_amd  # noqa: F401
    from evc_amd import sampler
    g = golden("traj_ddim_full")
    x_T, cond = rnd(831, 2, 15, 128, 128).cuda(), rnd(832, 2, 6, 128, 128).cuda()
    out = sampler.ddim_sampler(x_T, full_net, cond=cond, subsample_steps=5, denoise=True, clip_before=True,
                               final_only=True)[0].cpu()
    assert rel(out.reshape(2, -1)[:, ::30].numpy(), g["samples"]) < 5e-4
    assert rel(out[:, :, 0, :].numpy(), g["first_row"]) < 5e-4
    st = g["stats"]
    assert abs(float(out.std()) - float(st[1])) < 5e-4 * float(st[2])


def _psnr01(a, b):
    a, b = ((a + 1) / 2).clamp(0, 1).double(), ((b + 1) / 2).clamp(0, 1).double()
    return float(10 * torch.log10(1.0 / ((a - b) ** 2).mean()))


def test_full_size_full_length_ddpm_chunk_psnr_against_oracle(full_net, monkeypatch):
    """The benchmarked arithmetic over the REAL chain length: one full-size (ngf 192, 128x128, 262 M parameters) DDPM-100
    chunk = 100 ancestral steps + the denoise call = 101 chained score-network forwards (models/__init__.py:207-342 at
    configs/mine.yml:22), B = 1, injected x_T and per-step noise, HIP path under its default arithmetic (f16x3 convolutions
    and attention) against oracle/samplers.ddpm + oracle/scorenet.forward on the host cores.  Tolerance of SURVEY.md 8(c):
    PSNR >= 60 dB on the decoded [0, 1] frames; the range-event word must stay 0 (autouse fixture + explicit)."""
    import evc_amd  # noqa: F401
    from evc_amd import lib as L, sampler
    from oracle import samplers as OS, schedule as OSch, scorenet as ON
    torch.set_num_threads(16)
    d = ON.Dims()
    p = ON.seeded_params(d, 1234)
    x_T, cond = rnd(841, 1, 15, 128, 128), rnd(842, 1, 6, 128, 128).clamp(-1, 1)
    noises = [rnd(4000 + i, 1, 15, 128, 128) for i in range(100)]
    calls = {"n": 0}
    inner = full_net.forward_rows

    def counted(x, rows, cond=None):
        calls["n"] += 1
        return inner(x, rows, cond)
    monkeypatch.setattr(full_net, "forward_rows", counted)
    out = sampler.ddpm_sampler(x_T.cuda(), full_net, cond=cond.cuda(), subsample_steps=100, denoise=True, clip_before=True,
                               final_only=True, noise_fn=lambda i, x: noises[i])[0].cpu()
    assert calls["n"] == 101
    assert L.range_events() == 0
    ref = OS.ddpm(x_T.clone(), lambda x, t: ON.forward(p, d, x, t, cond=cond), OSch.base_schedule(),
                  subsample_steps=100, noise_fn=lambda i, x: noises[i])[0]
    psnr = _psnr01(out, ref)
    print(f"full-size DDPM-100 chunk (101 forwards): PSNR vs oracle {psnr:.1f} dB, max |diff| on [-1,1] {float((out - ref).abs().max()):.2e}")
    assert psnr >= 60.0, psnr


def test_full_size_full_length_fpndm50_chunk_psnr_against_oracle(full_net, monkeypatch):
    """BASELINE configs[4]'s chain at its real length: full-size F-PNDM with 50 subsampled steps = 3 Runge-Kutta warm-up
    iterations x 4 forwards + 47 = 59 chained forwards (models/__init__.py:39-100, models/pndm.py:3-52), B = 1, against the
    oracle sampler + oracle network on the host cores; PSNR >= 60 dB on the decoded [0, 1] frames, range events 0."""
    import evc_amd  # noqa: F401
    from evc_amd import lib as L, sampler
    from oracle import samplers as OS, schedule as OSch, scorenet as ON
    torch.set_num_threads(16)
    d = ON.Dims()
    p = ON.seeded_params(d, 1234)
    x_T, cond = rnd(851, 1, 15, 128, 128), rnd(852, 1, 6, 128, 128).clamp(-1, 1)
    calls = {"n": 0}
    inner = full_net.forward_rows

    def counted(x, rows, cond=None):
        calls["n"] += 1
        return inner(x, rows, cond)
    monkeypatch.setattr(full_net, "forward_rows", counted)
    out = sampler.FPNDM_sampler(x_T.cuda(), full_net, cond=cond.cuda(), subsample_steps=50, final_only=True,
                                clip_before=True)[0].cpu()
    assert calls["n"] == 59
    assert L.range_events() == 0
    ref = OS.fpndm(x_T.clone(), lambda x, t: ON.forward(p, d, x, t, cond=cond), OSch.base_schedule(), 50)
    ref = ref[0] if ref.dim() == 5 else ref
    psnr = _psnr01(out, ref)
    print(f"full-size F-PNDM-50 chunk (59 forwards): PSNR vs oracle {psnr:.1f} dB, max |diff| on [-1,1] {float((out - ref).abs().max()):.2e}")
    assert psnr >= 60.0, psnr


def test_sampler_trajectories_against_reference_goldens():
    import evc_amd  # noqa: F401
    from evc_amd import sampler
    g = golden("samplers_ngf32")
    net, d, p = build(32, 32, 32, 41)
    x_T, cond = rnd(42, 2, 15, 32, 32).cuda(), rnd(43, 2, 6, 32, 32).cuda()
    noises = [rnd(100 + i, 2, 15, 32, 32) for i in range(5)]
    out = sampler.ddpm_sampler(x_T, net, cond=cond, subsample_steps=5, denoise=True, clip_before=True,
                               final_only=True, noise_fn=lambda i, x: noises[i])
    assert out.shape == g["ddpm"].shape and rel(out, g["ddpm"]) < 5e-4
    out = sampler.ddim_sampler(x_T, net, cond=cond, subsample_steps=5, denoise=True, clip_before=True, final_only=True)
    assert rel(out, g["ddim"]) < 5e-4
    out = sampler.FPNDM_sampler(x_T, net, cond=cond, subsample_steps=10, final_only=True, clip_before=True)
    assert rel(out, g["fpndm"]) < 5e-4


def test_sampler_options_t_min_and_frac_steps_against_reference_goldens():
    """``t_min`` > 0 (start from a clean frame: skip the early steps, noise the input to the first executed level) and
    ``frac_steps`` (only the last fraction of the un-subsampled schedule) of the DDPM / DDIM loops -- options no shipped
    config sets (models/__init__.py:248-277, :145-157) -- against the reference run with the same injected noise;
    ``gamma`` on a model built without the Gamma buffers is NotImplementedError."""
    import evc_amd  # noqa: F401
    from evc_amd import sampler
    g = golden("sampler_options")
    net, d, p = build(32, 32, 32, 41)
    x0, cond = rnd(46, 2, 15, 32, 32).clamp(-1, 1).cuda(), rnd(43, 2, 6, 32, 32).cuda()

    def feed(n):
        it = iter([rnd(300 + i, 2, 15, 32, 32) for i in range(n)])
        return lambda tag, x: next(it)
    kw = dict(cond=cond, denoise=True, clip_before=True, final_only=True)
    out = sampler.ddpm_sampler(x0, net, subsample_steps=10, t_min=0.35, noise_fn=feed(int(g["ddpm_tmin_noises_used"])), **kw)
    assert out.shape == g["ddpm_tmin"].shape and rel(out, g["ddpm_tmin"]) < 5e-4
    out = sampler.ddim_sampler(x0, net, subsample_steps=10, t_min=0.35, noise_fn=feed(int(g["ddim_tmin_noises_used"])), **kw)
    assert rel(out, g["ddim_tmin"]) < 5e-4
    out = sampler.ddpm_sampler(x0, net, frac_steps=0.006, noise_fn=feed(int(g["ddpm_frac_noises_used"])), **kw)
    assert rel(out, g["ddpm_frac"]) < 5e-4
    with pytest.raises(NotImplementedError):  # gamma noise needs a model built with config.model.gamma (next test)
        sampler.ddpm_sampler(x0, net, subsample_steps=10, gamma=True, **kw)
    with pytest.raises(IndexError):           # the reference indexes the subsampled tables by label here, and fails the same way
        sampler.ddpm_sampler(x0, net, subsample_steps=10, frac_steps=0.5, **kw)


def test_sampler_gamma_noise_against_reference_goldens():
    """``gamma=True`` (Gamma-distributed noise, models/__init__.py:119-153, :225-278, :321-324) on a model built with
    ``config.model.gamma``: its k / k_cum / theta_t buffers equal to the reference's to 1 ulp (ncsnpp_more.py:744-749), and the
    DDPM / DDIM runs against the reference fed the same raw draws.  The raw draws are ~1e4-1e5 with a spread of order 1
    (fp32 quantises them at up to 0.016 -- in the reference too), so the centring must be the reference's fp32 subtraction."""
    import evc_amd  # noqa: F401
    from evc_amd import sampler
    from evc_amd.scorenet import ScoreNet
    from oracle.scorenet import Dims, seeded_params
    g = golden("sampler_gamma")
    cfg = make_config(32, 32, 32)
    cfg.model.gamma = True
    net = ScoreNet(cfg, seeded_params(Dims(ngf=32, n_head_channels=32, image_size=32), 41))
    for name in ("k", "k_cum", "theta_t"):     # 1 ulp: the host's vectorised fp32 sqrt / cumsum differ between CPU generations
        np.testing.assert_allclose(getattr(net, name).cpu().numpy(), g[name], rtol=5e-7, atol=0)
        # ... and k theta ~ 1e5 is subtracted from draws with a spread of order 1, so 1 ulp of theta moves the noise by
        # ~1 % (in the reference as well): the trajectories below run on the generating host's tables
        setattr(net, name, torch.from_numpy(g[name].copy()))
    x0, cond = rnd(46, 2, 15, 32, 32).clamp(-1, 1).cuda(), rnd(43, 2, 6, 32, 32).cuda()
    steps = list(range(0, 1000, 100))
    kw = dict(cond=cond, denoise=True, clip_before=True, final_only=True, subsample_steps=10, gamma=True)
    first = 0
    for name, fn, extra in (("ddpm_gamma", sampler.ddpm_sampler, {}), ("ddpm_gamma_tmin", sampler.ddpm_sampler, dict(t_min=0.35)),
                            ("ddim_gamma_tmin", sampler.ddim_sampler, dict(t_min=0.35))):
        feed, state, n = gamma_feed(g, name, first, net.k_cum.cpu(), net.theta_t.cpu(), steps)
        state["first_step"] = 1
        out = fn(x0, net, noise_fn=feed, **kw, **extra)
        assert state["n"] == n, name
        assert out.shape == g[name].shape and rel(out, g[name]) < 5e-4, name
        first += n
    # without injection the draw is torch's Gamma sampler on the device: finite, reproducible from the seed, and its
    # standardised noise has the moments the schedule promises (mean 0; variance k theta^2 / (1 - alpha) ~ 1)
    torch.manual_seed(5)
    a = sampler.ddpm_sampler(x0, net, **kw)
    torch.manual_seed(5)
    b = sampler.ddpm_sampler(x0, net, **kw)
    assert torch.isfinite(a).all() and torch.equal(a, b)
    i = 5
    z = sampler._gamma_noise(torch.empty(64, 15, 32, 32, device="cuda"), net.k_cum[steps[i]], net.theta_t[steps[i]],
                             net.alphas[steps[i]])
    assert abs(float(z.mean())) < 0.02 and abs(float(z.var()) - 1.0) < 0.1


def test_model_options_cond_emb_noise_in_cond_and_cosine_schedule():
    """UNetMore_DDPM options no shipped config sets, against the reference (ncsnpp_more.py:61,97-99,282-285,735-768):
    ``cond_emb`` (default mask and an explicit per-sample one), ``noise_in_cond`` (injected draw; every forward re-noises the
    conditioning frames), ``sigma_dist: cosine`` (buffers and a DDPM run); ``output_all_frames`` -- which fails in the reference
    itself -- stays NotImplementedError."""
    import evc_amd  # noqa: F401
    from evc_amd import sampler
    from evc_amd.scorenet import ScoreNet
    from oracle.scorenet import Dims, seeded_params
    g = golden("model_options")
    x, cond = rnd(61, 2, 15, 32, 32).cuda(), rnd(62, 2, 6, 32, 32).cuda()
    labels = torch.tensor([500, 7])

    def build_with(seed, **flags):
        cfg = make_config(32, 32, 32)
        for k, v in flags.items():
            setattr(cfg.model, k, v)
        d = Dims(ngf=32, n_head_channels=32, image_size=32, cond_emb=bool(flags.get("cond_emb", False)))
        return ScoreNet(cfg, seeded_params(d, seed))
    net = build_with(43, cond_emb=True)
    assert rel(net(x, labels, cond=cond), g["cond_emb_default"]) < 1e-4
    assert rel(net(x, labels, cond=cond, cond_mask=torch.tensor([1, 0], dtype=torch.int32)), g["cond_emb_mask10"]) < 1e-4
    assert rel(net.forward_label(x[:1], 500, cond[:1]), g["cond_emb_default"][:1]) < 1e-4       # the samplers' entry point
    net = build_with(44, noise_in_cond=True)
    calls = []
    net.cond_noise_fn = lambda c: (calls.append(tuple(c.shape)), rnd(63, 2, 6, 32, 32))[1]
    assert rel(net(x, labels, cond=cond), g["noise_in_cond"]) < 1e-4
    sampler.ddpm_sampler(x, net, cond=cond, subsample_steps=2, denoise=True, final_only=True, noise_fn=lambda t, xx: torch.zeros_like(xx))
    assert len(calls) == 1 + 3                                # one draw per forward: 2 steps + the denoise pass
    net.cond_noise_fn = None
    torch.manual_seed(3)
    a = net(x, labels, cond=cond)
    b = net(x, labels, cond=cond)
    assert torch.isfinite(a).all() and not torch.equal(a, b)  # fresh noise on every call
    with pytest.raises(IndexError):                           # alphas[labels] with a fractional label fails in the reference too
        net(x, torch.tensor([-0.5, -0.5]), cond=cond)
    net = build_with(45, sigma_dist="cosine")
    for name in ("betas", "alphas", "alphas_prev"):
        np.testing.assert_allclose(getattr(net, name).numpy(), g["cos_" + name], rtol=2e-6, atol=1e-9)
        setattr(net, name, torch.from_numpy(g["cos_" + name].copy()))          # cos() differs by an ulp between hosts
    it = iter([rnd(640 + i, 2, 15, 32, 32) for i in range(int(g["cos_noises_used"]))])
    out = sampler.ddpm_sampler(x, net, cond=cond, subsample_steps=10, denoise=True, clip_before=True, final_only=True,
                               noise_fn=lambda tag, xx: next(it))
    assert out.shape == g["cos_ddpm"].shape and rel(out, g["cos_ddpm"]) < 5e-4
    with pytest.raises(NotImplementedError):
        build_with(43, output_all_frames=True)
    with pytest.raises(NotImplementedError):
        build_with(43, sigma_dist="geometric")


def test_sampler_label_sequences_match_reference():
    import evc_amd  # noqa: F401
    from evc_amd import sampler
    g = golden("label_sequences")

    class Fake:
        def __init__(self):
            from oracle.schedule import base_schedule
            self.betas, self.alphas, self.alphas_prev = base_schedule()
            self.log = []

        def __call__(self, x, labels, cond=None):
            self.log.append(float(labels[0]))
            return 0.1 * x
    x = rnd(42, 2, 15, 32, 32).cuda()
    for key, fn, kw in (("labels_ddpm", sampler.ddpm_sampler, dict(subsample_steps=2, denoise=True)),
                        ("labels_ddim", sampler.ddim_sampler, dict(subsample_steps=4, denoise=True)),
                        ("labels_fpndm", sampler.FPNDM_sampler, dict(subsample_steps=4)),
                        ("labels_fpndm10", sampler.FPNDM_sampler, dict(subsample_steps=10)),
                        ("labels_ddpm100", sampler.ddpm_sampler, dict(subsample_steps=100, denoise=True))):
        f = Fake()
        fn(x, f, final_only=True, **kw)
        np.testing.assert_array_equal(np.asarray(f.log), g[key])


def test_hip_net_against_cpu_oracle_same_weights():
    """The oracle (pinned by the goldens) and the HIP path on fresh seeded inputs the goldens do not hold."""
    from oracle import scorenet as O
    net, d, p = build(32, 32, 32, 77)
    x, cond = rnd(78, 3, 15, 32, 32), rnd(79, 3, 6, 32, 32)
    ref = O.forward(p, d, x, torch.tensor([120, 120, 120]), cond=cond)
    out = net(x.cuda(), torch.tensor([120, 120, 120]), cond=cond.cuda())
    assert rel(out, ref.numpy()) < 1e-4
    # both placements of AdaGN + SiLU (one pre-activation pass per tensor / fused into the conv operand load)
    net.preactivate = not net.preactivate
    out2 = net(x.cuda(), torch.tensor([120, 120, 120]), cond=cond.cuda())
    assert rel(out2, ref.numpy()) < 1e-4 and rel(out2, out.cpu().numpy()) < 2e-5


def test_adversarial_adagn_rows_match_bf16x6_or_raise_the_range_flag():
    """A trained checkpoint could carry AdaGN (1 + scale) rows far larger than seeded ones.  With every row scaled up the
    fp16-split network must EITHER still agree with the exact bf16x6 network OR have raised EVC_RANGE_F16_OPERAND
    (and then its output may be anything, including NaN -- never silently wrong finite numbers with a clean flag)."""
    import os
    import evc_amd  # noqa: F401
    from evc_amd import lib as L
    from evc_amd.scorenet import ScoreNet
    from oracle.scorenet import Dims, seeded_params
    d = Dims(ngf=32, n_head_channels=32, image_size=32)
    p = seeded_params(d, 91)
    x, cond = rnd(92, 2, 15, 32, 32).cuda(), rnd(93, 2, 6, 32, 32).cuda()
    old = os.environ.get("EVC_CONV_ARITH")
    try:
        os.environ["EVC_CONV_ARITH"] = "f16x3"
        n16 = ScoreNet(make_config(32, 32, 32), p)
        os.environ["EVC_CONV_ARITH"] = "bf16x6"
        n6 = ScoreNet(make_config(32, 32, 32), p)
    finally:
        if old is None:
            os.environ.pop("EVC_CONV_ARITH", None)
        else:
            os.environ["EVC_CONV_ARITH"] = old
    for n in (n16, n6):
        n.prepare_labels([500.0])
    base16, base6 = n16._table.clone(), n6._table.clone()
    seen = set()
    for factor in (1.0, 50.0, 3000.0, 1e6):
        n16._table.copy_(base16 * factor)
        n6._table.copy_(base6 * factor)
        L.range_events(reset=True)
        o16 = n16.forward_label(x, 500, cond)
        ev = L.range_events(reset=True)
        o6 = n6.forward_label(x, 500, cond)
        L.range_events(reset=True)              # (the bf16x6 network shares the coefficient kernels: not its concern)
        seen.add(bool(ev & L.RANGE_F16_OPERAND))
        if not ev:
            assert bool(torch.isfinite(o16).all()) and rel(o16, o6.cpu().numpy()) < 2e-4, factor
        else:
            assert ev & L.RANGE_F16_OPERAND, (factor, ev)
    assert seen == {False, True}                # the seeded rows pass clean, the extreme ones are reported


def test_graph_replay_equals_eager_launches():
    """The captured HIP graph replays exactly the eager launch sequence (bitwise), across labels and repeats."""
    net, d, p = build(32, 32, 32, 55)
    x, cond = rnd(56, 2, 15, 32, 32).cuda(), rnd(57, 2, 6, 32, 32).cuda()
    net.use_graphs = False
    eager = [net.forward_label(x, lab, cond).clone() for lab in (0, 500, 990)]
    net.use_graphs = True
    for rep in range(2):
        for lab, ref in zip((0, 500, 990), eager):
            assert torch.equal(net.forward_label(x, lab, cond), ref)
    x2 = rnd(58, 2, 15, 32, 32).cuda()      # new inputs through the same graph
    net.use_graphs = False
    ref = net.forward_label(x2, 500, cond).clone()
    net.use_graphs = True
    assert torch.equal(net.forward_label(x2, 500, cond), ref)
    assert len(net._graphs) == 1      # one (stream, shape) -> one graph, reused for every label and input


def test_alt_model_unet_ddpm_against_reference_goldens():
    """SURVEY.md 8f item 4 -- the reference's other score network, models/unet.py::UNet_DDPM, behind the same sampler API:
    forward with / without time conditioning at two labels, a mixed-label batch, and a 4-step DDPM trajectory, against
    outputs of the imported reference (tests/golden/make_goldens.py::gen_unet_ddpm)."""
    import evc_amd  # noqa: F401
    from evc_amd import sampler
    from evc_amd.unet_ddpm import UNetDDPM
    from oracle import unet_ddpm as OU
    g = golden("unet_ddpm")
    x, cond = rnd(62, 2, 15, 32, 32).cuda(), rnd(63, 2, 6, 32, 32).cuda()
    for tc, tag in ((True, "tc"), (False, "notc")):
        cfg = make_config(32, 32, 32)
        cfg.model.time_conditional = tc
        net = UNetDDPM(cfg, OU.seeded_params(OU.Dims(ngf=32, time_conditional=tc), 61))
        for lab in (0, 500):
            assert rel(net(x, torch.tensor([lab, lab]), cond=cond), g[f"out_{tag}_t{lab}"]) < 1e-4, (tag, lab)
        if tc:
            mixed = net(x, torch.tensor([0, 500]), cond=cond)           # per-sample labels: one launch per distinct label
            assert rel(mixed[0], g["out_tc_t0"][0]) < 1e-4 and rel(mixed[1], g["out_tc_t500"][1]) < 1e-4
            np.testing.assert_array_equal(net.alphas.numpy(), g["alphas"])
            noises = [rnd(70 + i, 2, 15, 32, 32) for i in range(4)]
            out = sampler.ddpm_sampler(x, net, cond=cond, subsample_steps=4, denoise=True, clip_before=True,
                                       final_only=True, noise_fn=lambda i, xx: noises[i])
            assert out.shape == g["ddpm_tc"].shape and rel(out, g["ddpm_tc"]) < 5e-4
    cfg = make_config(64, 64, 32)                     # ngf 64: one 128-wide attention head (evc_attention_*, D = 128)
    cfg.model.time_conditional = True
    net = UNetDDPM(cfg, OU.seeded_params(OU.Dims(ngf=64, time_conditional=True), 64))
    out = net(rnd(62, 1, 15, 32, 32).cuda(), torch.tensor([500]), cond=rnd(63, 1, 6, 32, 32).cuda())
    assert rel(out, g["out_ngf64_t500"]) < 1e-4
    # the wrapper options (models/unet.py:337-372, shared with UNetMore_DDPM): noise_in_cond at mixed labels + cosine schedule
    cfg = make_config(32, 32, 32)
    cfg.model.time_conditional, cfg.model.noise_in_cond, cfg.model.sigma_dist = True, True, "cosine"
    net = UNetDDPM(cfg, OU.seeded_params(OU.Dims(ngf=32, time_conditional=True), 61))
    np.testing.assert_allclose(net.alphas.numpy(), g["cos_alphas"], rtol=2e-6, atol=1e-9)
    net.alphas = torch.from_numpy(g["cos_alphas"].copy())                    # cos() differs by an ulp between hosts
    net.cond_noise_fn = lambda c: rnd(64, 2, 6, 32, 32)
    assert rel(net(x, torch.tensor([500, 7]), cond=cond), g["out_nic_cos"]) < 1e-4


def build_spade(ngf, head, image_size, seed, spade_dim):
    import evc_amd  # noqa: F401
    from evc_amd.scorenet import build_score_network
    from oracle.scorenet import Dims
    from oracle.scorenet_spade import seeded_params
    d = Dims(ngf=ngf, n_head_channels=head, image_size=image_size)
    p = seeded_params(d, seed, spade_dim=spade_dim)
    cfg = make_config(ngf, head, image_size)
    cfg.model.spade = True
    cfg.model.spade_dim = spade_dim
    return build_score_network(cfg, p), d, p


def test_spade_scorenet_against_reference_golden():
    """SPADE-conditioned variant (reference SPADE_NCSNpp, ``model.spade: true``): integer, per-sample mixed and fractional
    labels against the imported reference's outputs; the per-chunk gamma/beta map cache follows the cond tensor."""
    from evc_amd.scorenet_spade import SpadeScoreNet
    g = golden("forward_spade")
    net, d, p = build_spade(32, 32, 32, 81, 32)
    assert isinstance(net, SpadeScoreNet)
    x, cond = rnd(82, 2, 15, 32, 32).cuda(), rnd(83, 2, 6, 32, 32).cuda()
    for key, lab in (("out_t0", [0, 0]), ("out_t990_3", [990, 3]), ("out_tm05", [-0.5, -0.5])):
        assert rel(net(x, torch.tensor(lab), cond=cond), g[key]) < 1e-4, key
    n_maps = len(net._maps)
    assert n_maps == 2 * sum(m["kind"] == "res" for m in net.program) + 1       # computed once, reused by the 3 forwards
    # another conditioning tensor: the maps are rebuilt (outputs change), and the first one gives the golden again
    other = net(x, torch.tensor([0, 0]), cond=cond.flip(0).contiguous())
    assert rel(other, g["out_t0"]) > 1e-2
    assert rel(net(x, torch.tensor([0, 0]), cond=cond), g["out_t0"]) < 1e-4
    # in-place update of the same tensor object (what a chunk loop that reuses its buffer does) is noticed too
    c2 = cond.clone()
    a = net(x, torch.tensor([0, 0]), cond=c2)
    c2.copy_(cond.flip(0))
    assert rel(net(x, torch.tensor([0, 0]), cond=c2), other.cpu().numpy()) < 1e-5 and rel(a, g["out_t0"]) < 1e-4
    with pytest.raises(ValueError):
        net(x, torch.tensor([0, 0]))


def test_spade_scorenet_through_the_ddpm_sampler_matches_oracle():
    """The SPADE network behind the same sampler API: 4 DDPM steps with injected noise against the CPU oracle."""
    import evc_amd  # noqa: F401
    from evc_amd import sampler as S
    from oracle import samplers as OS, schedule as OSch, scorenet_spade as OSP
    net, d, p = build_spade(32, 32, 32, 81, 32)
    x, cond = rnd(84, 2, 15, 32, 32), rnd(85, 2, 6, 32, 32)
    noises = [rnd(90 + i, 2, 15, 32, 32) for i in range(4)]
    ref = OS.ddpm(x.clone(), lambda xx, t: OSP.forward(p, d, xx, t, cond, spade_dim=32), OSch.base_schedule(),
                  subsample_steps=4, noise_fn=lambda i, xx: noises[i])
    out = S.ddpm_sampler(x.cuda(), net, cond=cond.cuda(), subsample_steps=4, denoise=True, clip_before=True,
                         final_only=True, noise_fn=lambda i, xx: noises[i])[0].cpu()
    assert rel(out, ref.numpy()) < 2e-4


def test_spade_scorenet_full_size_against_oracle():
    """The SPADE variant at the full configuration (ngf 192, 128x128, spade_dim 128; 5 resolution levels, every tile /
    split choice of the real layer shapes) against the CPU oracle -- itself pinned to the reference at reduced size --
    on the same seeded weights, mixed labels in one batch."""
    torch.set_num_threads(16)
    from oracle import scorenet_spade as OSP
    net, d, p = build_spade(192, 192, 128, 91, 128)
    x, cond = rnd(92, 2, 15, 128, 128), rnd(93, 2, 6, 128, 128)
    labels = torch.tensor([700, 20])
    ref = OSP.forward(p, d, x, labels, cond, spade_dim=128)
    out = net(x.cuda(), labels, cond=cond.cuda())
    assert rel(out, ref.numpy()) < 2e-4


ARCHS_3D = [("unetmorepseudo3d", "forward_pseudo3d", 91), ("unetmore3d", "forward_conv3d", 96)]


def build_pseudo3d(arch="unetmorepseudo3d", seed=91):
    import evc_amd  # noqa: F401
    from evc_amd.scorenet import build_score_network
    from oracle import scorenet_pseudo3d as O3
    from test_oracle_goldens import pseudo3d_dims
    d = pseudo3d_dims()
    cfg = make_config(d.ngf, d.n_head_channels, d.image_size)
    cfg.model.arch, cfg.model.ch_mult, cfg.model.num_res_blocks = arch, d.ch_mult, d.num_res_blocks
    cfg.model.attn_resolutions = d.attn_resolutions
    cfg.data.num_frames, cfg.data.num_frames_cond = d.num_frames, d.num_frames_cond
    p = O3.seeded_params(d, seed, arch=arch)
    return build_score_network(cfg, p), d, p, O3


@pytest.mark.parametrize("arch,name,seed", ARCHS_3D)
def test_pseudo3d_network_against_reference_goldens(arch, name, seed):
    """``arch: unetmorepseudo3d`` / ``unetmore3d`` (ncsnpp_more.py is3d / pseudo3d branches + models/better/layers3d.py) on the
    HIP kernels: outputs at integer, mixed and fractional labels and the output of EVERY module of ``all_modules`` -- pseudo-3-D
    (or Conv3d) convolutions, res-blocks with the 3-D AdaGN, space + time attention blocks, the frame converters -- against the
    reference's own (tests/golden/make_goldens.py::gen_forward_pseudo3d, forward hooks), fp32 tolerance 1e-4."""
    from evc_amd.scorenet_pseudo3d import Conv3dScoreNet, Pseudo3dScoreNet
    g = golden(name)
    net, d, p, O3 = build_pseudo3d(arch, seed)
    assert type(net) is (Conv3dScoreNet if arch == "unetmore3d" else Pseudo3dScoreNet)
    x, cond = rnd(92, 2, 9, 16, 16).cuda(), rnd(93, 2, 6, 16, 16).cuda()
    for key, lab in (("out_t0", [0, 0]), ("out_tm05", [-0.5, -0.5])):
        out = net(x, torch.tensor(lab), cond=cond)
        assert out.shape == (2, 9, 16, 16)
        assert rel(out, g[key]) < 1e-4, key
    net.taps = {}
    try:
        out = net(x, torch.tensor([430, 7]), cond=cond)
        taps = net.taps
    finally:
        net.taps = None
    assert rel(out, g["out_t430_7"]) < 1e-4
    kinds, worst = {}, 0.0
    for idx, m in enumerate(O3.program(d)):
        if m["kind"] in ("linear", "norm"):
            continue
        t = taps[idx].float().cpu()
        assert tuple(t.shape) == tuple(g[f"tapshape{idx}"]), (idx, m)
        flat = t.reshape(-1)
        stride = max(1, flat.numel() // 512)
        err = rel(flat[::stride][:512].numpy(), g[f"tap{idx}"])
        worst = max(worst, err)
        assert err < 1e-4, (idx, m, err)
        mean, std = g[f"tapstat{idx}"]
        assert abs(float(t.std()) - float(std)) < 1e-4 * float(std), (idx, m)
        kinds[m["kind"]] = kinds.get(m["kind"], 0) + 1
    print(f"{arch}: {sum(kinds.values())} module outputs checked, worst relative error {worst:.2e}")
    assert kinds == dict(conv3=2, res=10, attn=5, mix=5), kinds


@pytest.mark.parametrize("arch,name,seed", ARCHS_3D)
def test_pseudo3d_network_in_the_sampler_against_the_oracle(arch, name, seed):
    """The pseudo-3-D (or Conv3d) network plugs into the same sampling loop: a 10-step DDPM chunk (11 forwards) with injected noise
    against oracle/samplers.ddpm over oracle/scorenet_pseudo3d.forward (same weights, same draws)."""
    from evc_amd import sampler
    from oracle import samplers as OS, schedule as OSch
    net, d, p, O3 = build_pseudo3d(arch, seed)
    B = 2
    xT, cond = rnd(94, B, 9, 16, 16), rnd(95, B, 6, 16, 16).clamp(-1, 1)
    noises = [rnd(200 + k, B, 9, 16, 16) for k in range(10)]
    out = sampler.ddpm_sampler(xT.cuda(), net, cond=cond.cuda(), subsample_steps=10, denoise=True, clip_before=True,
                               final_only=True, noise_fn=lambda i, x: noises[i])[0].cpu()
    ref = OS.ddpm(xT.clone(), lambda x, t: O3.forward(p, d, x, t, cond=cond, arch=arch), OSch.base_schedule(), subsample_steps=10,
                  noise_fn=lambda i, x: noises[i])[0]
    assert rel(out, ref.numpy()) < 2e-4


@pytest.mark.parametrize("arch", ["unetmorepseudo3d", "unetmore3d"])
def test_3d_networks_at_the_shipped_frame_counts_against_the_oracle(arch):
    """The 3-D archs at the shipped clip geometry -- 5 generated + 2 conditioning frames, 64 x 64, three levels (attention at
    16 x 16, head width 64) -- so that the frame axis is 7 -> 5, the time convolution's "image" is 7 rows x 4096 columns and the
    3-D GroupNorm runs over 7 x 64 x 64 pixels; against the pinned oracle (oracle/scorenet_pseudo3d.py) on seeded weights."""
    import evc_amd  # noqa: F401
    from evc_amd import synthetic
    from evc_amd.scorenet import build_score_network
    from oracle import scorenet_pseudo3d as O3
    torch.set_num_threads(16)
    d = O3.Dims(ngf=64, ch_mult=[1, 2, 2], num_res_blocks=1, attn_resolutions=[16], n_head_channels=64, image_size=64,
                num_frames=5, num_frames_cond=2)
    cfg = make_config(d.ngf, d.n_head_channels, d.image_size)
    cfg.model.arch, cfg.model.ch_mult, cfg.model.num_res_blocks, cfg.model.attn_resolutions = arch, d.ch_mult, 1, d.attn_resolutions
    p = synthetic.diffusion_state_dict(cfg, 57)
    assert [k for k, _ in O3.param_shapes(d, arch=arch)] == list(p)
    net = build_score_network(cfg, p)
    x, cond = rnd(58, 2, 15, 64, 64), rnd(59, 2, 6, 64, 64)
    out = net(x.cuda(), torch.tensor([321, 5]), cond=cond.cuda())
    ref = O3.forward(p, d, x, torch.tensor([321, 5]), cond=cond, arch=arch)
    assert out.shape == (2, 15, 64, 64)
    assert rel(out, ref.numpy()) < 1e-4


@pytest.mark.parametrize("arch,name,seed", ARCHS_3D)
def test_3d_networks_replayed_from_hip_graphs_equal_eager(arch, name, seed):
    """``use_graphs=True`` (forwards replayed from a captured HIP graph, ``bench.py --graphs``) with the 3-D networks: their host
    code allocates and repeats coefficient rows inside the capture; outputs must equal the eager launches bit for bit, also on
    a second input through the same graph."""
    eager, d, p, O3 = build_pseudo3d(arch, seed)
    import copy
    from evc_amd.scorenet import build_score_network
    graph = build_score_network(copy.deepcopy(eager.config), p, use_graphs=True)
    x, cond = rnd(92, 2, 9, 16, 16).cuda(), rnd(93, 2, 6, 16, 16).cuda()
    lab = torch.tensor([430, 7])
    assert torch.equal(graph(x, lab, cond=cond), eager(x, lab, cond=cond))
    assert torch.equal(graph(0.5 * x, lab, cond=cond), eager(0.5 * x, lab, cond=cond))
