"""The CPU oracle (oracle/) checked against fixtures produced by the reference itself
(tests/golden/make_goldens.py).  This is what pins the oracle; nothing here touches the GPU."""
import numpy as np
import pytest
import torch

from conftest import gamma_feed, golden, rnd
from oracle import samplers, schedule, scorenet, upfirdn2d
from oracle.scorenet import Dims, seeded_params


def test_schedule_tables():
    g = golden("schedule")
    betas, alphas, alphas_prev = schedule.base_schedule()
    np.testing.assert_array_equal(betas.numpy(), g["betas"])
    np.testing.assert_array_equal(alphas.numpy(), g["alphas"])
    np.testing.assert_array_equal(alphas_prev.numpy(), g["alphas_prev"])
    for S in (10, 50, 100):
        steps, a, ap, b = schedule.subsample(alphas, alphas_prev, betas, S)
        np.testing.assert_array_equal(steps.numpy(), g[f"steps_{S}"])
        np.testing.assert_array_equal(a.numpy(), g[f"alphas_{S}"])
        np.testing.assert_array_equal(ap.numpy(), g[f"alphas_prev_{S}"])
        np.testing.assert_array_equal(b.numpy(), g[f"betas_{S}"])


def test_fir_resampling_numpy_and_torch_forms():
    g = golden("fir")
    x = g["x"]
    np.testing.assert_allclose(upfirdn2d.upsample_2d(x), g["up"], atol=2e-6)
    np.testing.assert_allclose(upfirdn2d.downsample_2d(x), g["down"], atol=2e-6)
    xt = torch.from_numpy(x)
    np.testing.assert_allclose(scorenet.fir_up2(xt).numpy(), g["up"], atol=2e-6)
    np.testing.assert_allclose(scorenet.fir_down2(xt).numpy(), g["down"], atol=2e-6)


def test_upfirdn2d_generic():
    g = golden("upfirdn2d_generic")
    x, k = g["x"], g["k"]
    np.testing.assert_allclose(upfirdn2d.upfirdn2d(x, k, up=2, down=1, pad=(1, 0)), g["up2_pad10"], atol=3e-6)
    np.testing.assert_allclose(upfirdn2d.upfirdn2d(x, k, up=1, down=3, pad=(2, 1)), g["down3_pad21"], atol=3e-6)
    np.testing.assert_allclose(upfirdn2d.upfirdn2d(x, k, up=3, down=2, pad=(2, 2)), g["up3_down2_pad22"], atol=3e-6)


def _rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def test_blocks_every_module_output():
    g = golden("blocks_ngf64")
    d = Dims(ngf=64, n_head_channels=64, image_size=32)
    p = seeded_params(d, 21)
    x, cond = rnd(22, 2, 15, 32, 32), rnd(23, 2, 6, 32, 32)
    taps = {}
    out = scorenet.forward(p, d, x, torch.tensor([430, 430]), cond=cond, taps=taps)
    assert _rel(out.numpy(), g["out"]) < 2e-5
    mods = scorenet.program(d)
    checked = 0
    for idx, m in enumerate(mods):
        if m["kind"] not in ("res", "attn", "conv3") or idx not in taps:
            continue
        t = taps[idx].reshape(-1)
        stride = max(1, t.numel() // 512)
        assert _rel(t[::stride][:512].numpy(), g[f"tap{idx}"]) < 2e-5, (idx, m)
        checked += 1
    assert checked >= 40


def test_forward_reduced_labels_incl_fractional():
    g = golden("forward_ngf32")
    d = Dims(ngf=32, n_head_channels=32, image_size=32)
    p = seeded_params(d, 31)
    x, cond = rnd(32, 2, 15, 32, 32), rnd(33, 2, 6, 32, 32)
    for key, lab in (("out_t0", [0, 0]), ("out_t990", [990, 990]), ("out_tm05", [-0.5, -0.5])):
        out = scorenet.forward(p, d, x, torch.tensor(lab), cond=cond)
        assert _rel(out.numpy(), g[key]) < 2e-5, key


def _net(seed):
    d = Dims(ngf=32, n_head_channels=32, image_size=32)
    p = seeded_params(d, seed)
    return d, p


def test_sampler_trajectories():
    g = golden("samplers_ngf32")
    d, p = _net(41)
    x_T, cond = rnd(42, 2, 15, 32, 32), rnd(43, 2, 6, 32, 32)
    eps = lambda x, t: scorenet.forward(p, d, x, t, cond=cond)
    sched = schedule.base_schedule()
    noises = [rnd(100 + i, 2, 15, 32, 32) for i in range(5)]
    out = samplers.ddpm(x_T.clone(), eps, sched, subsample_steps=5, noise_fn=lambda i, x: noises[i])
    assert out.shape == g["ddpm"].shape and _rel(out.numpy(), g["ddpm"]) < 1e-4
    out = samplers.ddim(x_T.clone(), eps, sched, subsample_steps=5)
    assert _rel(out.numpy(), g["ddim"]) < 1e-4
    out = samplers.fpndm(x_T.clone(), eps, sched, subsample_steps=10)
    assert _rel(out.numpy(), g["fpndm"]) < 1e-4


def test_sampler_options_t_min_and_frac_steps():
    """The DDPM / DDIM options no shipped config sets, against the reference run with injected noise
    (models/__init__.py:248-277, :145-157): t_min > 0 and frac_steps."""
    g = golden("sampler_options")
    d, p = _net(41)
    x0, cond = rnd(46, 2, 15, 32, 32).clamp(-1, 1), rnd(43, 2, 6, 32, 32)
    eps = lambda x, t: scorenet.forward(p, d, x, t, cond=cond)
    sched = schedule.base_schedule()

    def feed(n):
        it = iter([rnd(300 + i, 2, 15, 32, 32) for i in range(n)])
        return lambda tag, x: next(it)
    out = samplers.ddpm(x0.clone(), eps, sched, subsample_steps=10, t_min=0.35, noise_fn=feed(int(g["ddpm_tmin_noises_used"])))
    assert _rel(out.numpy(), g["ddpm_tmin"]) < 1e-4
    out = samplers.ddim(x0.clone(), eps, sched, subsample_steps=10, t_min=0.35, noise_fn=feed(int(g["ddim_tmin_noises_used"])))
    assert _rel(out.numpy(), g["ddim_tmin"]) < 1e-4
    out = samplers.ddpm(x0.clone(), eps, sched, frac_steps=0.006, noise_fn=feed(int(g["ddpm_frac_noises_used"])))
    assert _rel(out.numpy(), g["ddpm_frac"]) < 1e-4


def test_sampler_gamma_noise_against_reference_goldens():
    """``gamma=True`` of the DDPM / DDIM loops (models/__init__.py:119-153, :225-278, :321-324) and the model buffers it
    reads (ncsnpp_more.py:744-749), against the reference run on a ``config.model.gamma`` model with the same raw draws."""
    g = golden("sampler_gamma")
    sched = schedule.base_schedule()
    k, k_cum, theta_t = schedule.gamma_schedule(sched[0], sched[1])
    for name, v in (("k", k), ("k_cum", k_cum), ("theta_t", theta_t)):       # bit-equal on the generating host; 1 ulp across CPUs
        np.testing.assert_allclose(v.numpy(), g[name], rtol=5e-7, atol=0)
    # k theta ~ 1e5 is subtracted from draws with a spread of order 1: 1 ulp of theta moves the noise by ~1 % (in the
    # reference as well), so the trajectories run on the generating host's tables
    k_cum, theta_t = torch.from_numpy(g["k_cum"].copy()), torch.from_numpy(g["theta_t"].copy())
    d, p = _net(41)
    x0, cond = rnd(46, 2, 15, 32, 32).clamp(-1, 1), rnd(43, 2, 6, 32, 32)
    eps = lambda x, t: scorenet.forward(p, d, x, t, cond=cond)
    steps = list(range(0, 1000, 100))
    first_tmin = next(i for i, st in enumerate(steps) if not st < 0.35 * 10)     # len(alphas) is the SUBSAMPLED length (:145, :263)
    first = 0
    for name, fn, kw in (("ddpm_gamma", samplers.ddpm, {}), ("ddpm_gamma_tmin", samplers.ddpm, dict(t_min=0.35)),
                         ("ddim_gamma_tmin", samplers.ddim, dict(t_min=0.35))):
        feed, state, n = gamma_feed(g, name, first, k_cum, theta_t, steps)
        state["first_step"] = first_tmin
        out = fn(x0.clone(), eps, sched, subsample_steps=10, noise_fn=feed, gamma=(k_cum, theta_t), **kw)
        assert state["n"] == n
        assert _rel(out.numpy(), g[name]) < 1e-4, name
        first += n


def test_model_options_cond_emb_noise_in_cond_and_cosine_schedule():
    """UNetMore_DDPM options no shipped config sets, against the reference (ncsnpp_more.py:61,97-99,282-285,735-768): the
    cond_emb embedding row appended to the time embedding (default mask and an explicit one), noise_in_cond with the injected
    draw, the cosine schedule's buffers and a DDPM run on them."""
    g = golden("model_options")
    x, cond = rnd(61, 2, 15, 32, 32), rnd(62, 2, 6, 32, 32)
    labels = torch.tensor([500, 7])
    d = Dims(ngf=32, n_head_channels=32, image_size=32, cond_emb=True)
    p = seeded_params(d, 43)
    assert _rel(scorenet.forward(p, d, x, labels, cond=cond).numpy(), g["cond_emb_default"]) < 1e-4
    out = scorenet.forward(p, d, x, labels, cond=cond, cond_mask=torch.tensor([1, 0], dtype=torch.int32))
    assert _rel(out.numpy(), g["cond_emb_mask10"]) < 1e-4
    assert _rel(g["cond_emb_mask10"][0], g["cond_emb_default"][0]) < 1e-6 and _rel(g["cond_emb_mask10"][1], g["cond_emb_default"][1]) > 1e-3
    d, p = _net(44)
    sched = schedule.base_schedule()
    noised = scorenet.noise_cond(cond, labels, sched[1], rnd(63, 2, 6, 32, 32))
    assert _rel(scorenet.forward(p, d, x, labels, cond=noised).numpy(), g["noise_in_cond"]) < 1e-4
    cos = schedule.cosine_schedule()
    for name, v in zip(("cos_betas", "cos_alphas", "cos_alphas_prev"), cos):
        np.testing.assert_allclose(v.numpy(), g[name], rtol=2e-6, atol=1e-9)      # cos() differs by an ulp between hosts
    cos = tuple(torch.from_numpy(g[k].copy()) for k in ("cos_betas", "cos_alphas", "cos_alphas_prev"))
    d, p = _net(45)
    it = iter([rnd(640 + i, 2, 15, 32, 32) for i in range(int(g["cos_noises_used"]))])
    out = samplers.ddpm(x.clone(), lambda xx, t: scorenet.forward(p, d, xx, t, cond=cond), cos, subsample_steps=10,
                        noise_fn=lambda tag, xx: next(it))
    assert _rel(out.numpy(), g["cos_ddpm"]) < 1e-4


def test_label_sequences():
    g = golden("label_sequences")
    sched = schedule.base_schedule()
    x = rnd(42, 2, 15, 32, 32)

    def run(fn, **kw):
        log = []

        def eps(xx, t):
            log.append(float(t[0]))
            return 0.1 * xx
        fn(x.clone(), eps, sched, **kw)
        return np.asarray(log)
    np.testing.assert_array_equal(run(samplers.ddpm, subsample_steps=2, noise_fn=lambda i, t: torch.zeros_like(t)),
                                  g["labels_ddpm"])
    np.testing.assert_array_equal(run(samplers.ddim, subsample_steps=4), g["labels_ddim"])
    np.testing.assert_array_equal(run(samplers.fpndm, subsample_steps=4), g["labels_fpndm"])
    np.testing.assert_array_equal(run(samplers.fpndm, subsample_steps=10), g["labels_fpndm10"])
    lab = run(samplers.ddpm, subsample_steps=100, noise_fn=lambda i, t: torch.zeros_like(t))
    np.testing.assert_array_equal(lab, g["labels_ddpm100"])
    assert len(lab) == 101 and lab[-1] == 99  # the L-1 denoise-label quirk (models/__init__.py:333-335)


@pytest.mark.slow
def test_forward_full_size():
    g = golden("forward_full")
    d = Dims()
    p = seeded_params(d, 1234)
    x, cond = rnd(51, 1, 15, 128, 128), rnd(52, 1, 6, 128, 128)
    o = scorenet.forward(p, d, x, torch.tensor([500]), cond=cond)
    assert _rel(o.reshape(-1)[::60].numpy(), g["samples"]) < 5e-5
    assert _rel(o[0, :, 0, :].numpy(), g["first_row"]) < 5e-5


def test_unet_ddpm_oracle_matches_reference_goldens():
    """The alternative score network (reference models/unet.py::UNet_DDPM): oracle forward with and without time
    conditioning, and a 4-step DDPM trajectory through the oracle sampler, against outputs of the imported reference."""
    from oracle import samplers as OS, schedule as OSch, unet_ddpm as OU
    g = golden("unet_ddpm")
    x, cond = rnd(62, 2, 15, 32, 32), rnd(63, 2, 6, 32, 32)
    for tc, tag in ((True, "tc"), (False, "notc")):
        d = OU.Dims(ngf=32, time_conditional=tc)
        p = OU.seeded_params(d, 61)
        for lab in (0, 500):
            out = OU.forward(p, d, x, torch.tensor([lab, lab]), cond=cond)
            ref = g[f"out_{tag}_t{lab}"]
            assert float(np.abs(out.numpy() - ref).max() / np.abs(ref).max()) < 2e-5, (tag, lab)
    assert not np.array_equal(g["out_tc_t0"], g["out_tc_t500"])          # the label matters only when conditional
    assert np.array_equal(g["out_notc_t0"], g["out_notc_t500"])
    d = OU.Dims(ngf=32, time_conditional=True)
    p = OU.seeded_params(d, 61)
    noises = [rnd(70 + i, 2, 15, 32, 32) for i in range(4)]
    traj = OS.ddpm(x.clone(), lambda xx, t: OU.forward(p, d, xx, t, cond=cond), OSch.base_schedule(), subsample_steps=4,
                   noise_fn=lambda i, xx: noises[i])
    assert float(np.abs(traj.numpy() - g["ddpm_tc"]).max() / np.abs(g["ddpm_tc"]).max()) < 1e-4
    d = OU.Dims(ngf=64, time_conditional=True)                          # 128-wide attention head
    out = OU.forward(OU.seeded_params(d, 64), d, rnd(62, 1, 15, 32, 32), torch.tensor([500]), cond=rnd(63, 1, 6, 32, 32))
    assert float(np.abs(out.numpy() - g["out_ngf64_t500"]).max() / np.abs(g["out_ngf64_t500"]).max()) < 2e-5


def test_spade_scorenet_oracle_matches_reference_goldens():
    """SPADE-conditioned NCSN++ (``model.spade: true``, reference SPADE_NCSNpp): oracle forward at integer, mixed and
    fractional labels against outputs of the imported reference on the same seeded weights."""
    from oracle import scorenet as OSN, scorenet_spade as OSP
    g = golden("forward_spade")
    d = OSN.Dims(ngf=32, n_head_channels=32, image_size=32)
    p = OSP.seeded_params(d, 81, spade_dim=32)
    x, cond = rnd(82, 2, 15, 32, 32), rnd(83, 2, 6, 32, 32)
    for key, labels in (("out_t0", [0, 0]), ("out_t990_3", [990, 3]), ("out_tm05", [-0.5, -0.5])):
        out = OSP.forward(p, d, x, torch.tensor(labels), cond, spade_dim=32)
        ref = g[key]
        assert float(np.abs(out.numpy() - ref).max() / np.abs(ref).max()) < 2e-5, key
    assert not np.array_equal(g["out_t0"], g["out_t990_3"])


def pseudo3d_dims():
    """The reduced ``unetmorepseudo3d`` configuration of tests/golden/make_goldens.py::gen_forward_pseudo3d."""
    return Dims(ngf=32, ch_mult=[1, 2], num_res_blocks=1, attn_resolutions=[16, 8], n_head_channels=32, image_size=16,
                channels=3, num_frames=3, num_frames_cond=2)


@pytest.mark.parametrize("arch,name,seed", [("unetmorepseudo3d", "forward_pseudo3d", 91), ("unetmore3d", "forward_conv3d", 96)])
def test_pseudo3d_scorenet_oracle_matches_reference_goldens(arch, name, seed):
    """``model.arch: unetmorepseudo3d`` / ``unetmore3d`` (ncsnpp_more.py is3d / pseudo3d branches + models/better/layers3d.py):
    the oracle's forward at integer, mixed and fractional labels and the output of every module of ``all_modules`` (pseudo-3-D
    or Conv3d convolutions, 3-D AdaGN res-blocks, space + time attention, frame converters) against the reference's own."""
    from oracle import scorenet_pseudo3d as O3
    g = golden(name)
    d = pseudo3d_dims()
    p = O3.seeded_params(d, seed, arch=arch)
    x, cond = rnd(92, 2, 9, 16, 16), rnd(93, 2, 6, 16, 16)
    for key, lab in (("out_t0", [0, 0]), ("out_tm05", [-0.5, -0.5])):
        assert _rel(O3.forward(p, d, x, torch.tensor(lab), cond=cond, arch=arch).numpy(), g[key]) < 2e-5, key
    taps = {}
    out = O3.forward(p, d, x, torch.tensor([430, 7]), cond=cond, taps=taps, arch=arch)
    assert _rel(out.numpy(), g["out_t430_7"]) < 2e-5
    mods = O3.program(d)
    kinds = {}
    for idx, m in enumerate(mods):
        if m["kind"] in ("linear", "norm"):
            continue
        t = taps[idx]
        assert tuple(t.shape) == tuple(g[f"tapshape{idx}"]), (idx, m)
        flat = t.reshape(-1)
        stride = max(1, flat.numel() // 512)
        assert _rel(flat[::stride][:512].numpy(), g[f"tap{idx}"]) < 2e-5, (idx, m)
        kinds[m["kind"]] = kinds.get(m["kind"], 0) + 1
    assert kinds == dict(conv3=2, res=10, attn=5, mix=5), kinds
