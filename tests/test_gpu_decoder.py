"""GPU test of the receiver loop (evc_amd/decoder.py): transmit-mask semantics, batched key-frame runs, and the
generated frames against the CPU oracle sampler fed the same conditioning frames and injected noise."""
import numpy as np
import pytest
import torch

from conftest import rnd

pytestmark = pytest.mark.gpu


def test_clip_decoder_mask_semantics_and_generated_frames_vs_oracle():
    import evc_amd  # noqa: F401
    from evc_amd import sampler as S, synthetic
    from evc_amd.config import default_config
    from evc_amd.decoder import ClipDecoder, total_bits
    from evc_amd.elic import ElicModel, count_bits
    from evc_amd.scorenet import ScoreNet
    from oracle import samplers as OS, schedule as OSch, scorenet as ON

    cfg = default_config(32, 32, 64, subsample=3)
    d_net = ON.Dims(ngf=32, n_head_channels=32, image_size=64)
    p = ON.seeded_params(d_net, 9)
    net = ScoreNet(cfg, p)
    elic = ElicModel(synthetic.elic_state_dict(4))
    dec = ClipDecoder(net, elic, cfg, S.get_sampler("DDPM"))

    B, F = 2, 12
    clips = torch.from_numpy(synthetic.make_clips(B, seed=5, frames=F, size=64).astype(np.float32) / 255)
    mask = np.array([1, 1, 0, 0, 0, 0, 0, 0, 0, 1, 1, 0])
    key_pos = [0, 1, 9, 10]
    key_strings, shape = [], None
    for f in key_pos:
        enc = elic.compress(clips[:, f].cuda())
        key_strings.append(enc["strings"])
        shape = enc["shape"]
    assert total_bits(key_strings) == sum(count_bits(s) for s in key_strings) > 0

    noise_log = []

    def noise_fn(tag, shp):
        t = rnd(1000 + len(noise_log), *shp)
        noise_log.append((tag, t))
        return t.cuda()
    out = dec.decode(mask, key_strings, shape, frames=F, noise_fn=noise_fn)
    assert out.shape == (B, F, 3, 64, 64) and float(out.min()) >= 0 and float(out.max()) <= 1
    # key frames are exactly what the codec decodes for each position (batched runs == single decodes)
    for k, f in enumerate(key_pos):
        one = elic.decompress(key_strings[k], shape)["x_hat"]
        assert torch.equal(out[:, f], one)
    # three generation calls happened: frames 2-6 (5 kept), 7-8 (2 of 5 kept), 11 (1 of 5 kept)
    inits = [i for i, (tag, _) in enumerate(noise_log) if tag == "init"]
    assert len(inits) == 3
    # oracle: regenerate chunk 1 and chunk 2 from the decoded frames with the same noise
    sched = OSch.base_schedule()
    outc = out.cpu()

    def oracle_chunk(prev2, log):
        cond = (2 * prev2.reshape(B, 6, 64, 64) - 1)
        x_T = log[0][1]
        steps = {tag: t for tag, t in log[1:]}
        x = OS.ddpm(x_T.clone(), lambda x, t: ON.forward(p, d_net, x, t, cond=cond), sched, subsample_steps=3,
                    noise_fn=lambda i, x: steps[i])
        return ((x[0] + 1) / 2).clamp(0, 1).reshape(B, 5, 3, 64, 64)
    g1 = oracle_chunk(outc[:, 0:2], noise_log[inits[0]:inits[1]])
    assert float((outc[:, 2:7] - g1).abs().max()) < 2e-3            # fp32 sampler tolerance on [0,1] pixels
    g2 = oracle_chunk(outc[:, 5:7], noise_log[inits[1]:inits[2]])
    assert float((outc[:, 7:9] - g2[:, :2]).abs().max()) < 2e-3     # only as many frames as the mask asks for
    g3 = oracle_chunk(outc[:, 9:11], noise_log[inits[2]:])
    assert float((outc[:, 11:12] - g3[:, :1]).abs().max()) < 2e-3


def test_generation_is_invariant_to_concurrent_clip_grouping():
    """Clips are independent: sampling them as 1 batch or as 3 concurrent groups (one HIP stream each) gives the
    same frames (up to the split-K summation order, which depends on the batch size of a launch)."""
    import evc_amd  # noqa: F401
    from evc_amd import sampler as S
    from evc_amd.config import default_config
    from evc_amd.decoder import ClipDecoder
    from evc_amd.scorenet import ScoreNet
    from oracle import scorenet as ON
    cfg = default_config(32, 32, 32, subsample=4)
    net = ScoreNet(cfg, ON.seeded_params(ON.Dims(ngf=32, n_head_channels=32, image_size=32), 12))
    cond = rnd(60, 5, 2, 3, 32, 32).clamp(-1, 1).add(1).div(2).cuda()

    def noise_fn(tag, shp):      # pure function of the tag, so every group slices the same tensor
        return rnd(hash(str(tag)) % 1000 + 7, *shp)
    # F-PNDM: its Runge-Kutta midpoint / -1 label rows must be in the AdaGN table before the groups fan out to their
    # own streams (a row built lazily on one group's stream was read by the others with no ordering)
    for name, graphs in (("DDPM", False), ("DDIM", False), ("FPNDM", False), ("DDPM", True), ("FPNDM", True)):
        net.use_graphs = graphs
        net._rows.clear(); net._row_tensors.clear(); net._n_rows = 0; net._graphs.clear()    # every sampler starts cold
        dec = ClipDecoder(net, None, cfg, S.get_sampler(name))
        one = dec.generate(cond, noise_fn=noise_fn, groups=1)
        three = dec.generate(cond, noise_fn=noise_fn, groups=3)
        assert one.shape == three.shape == (5, 5, 3, 32, 32)
        assert float((one - three).abs().max()) < 1e-3
    # without injected noise the grouped path draws from per-group Philox generators and still returns valid frames
    dec = ClipDecoder(net, None, cfg, S.get_sampler("DDPM"), groups=2)
    out = dec.generate(cond, generator=torch.Generator(device="cuda").manual_seed(3))
    assert out.shape == (5, 5, 3, 32, 32) and bool(torch.isfinite(out).all())


def test_full_ddpm_chunk_psnr_against_oracle():
    """One whole chunk with the reference's schedule (1000 steps subsampled to 100 + the denoise call = 101 network
    evaluations, clip to [-1, 1] every step) on a reduced network: decoded [0,1] frames of the HIP path vs the CPU
    oracle with the same injected noise.  Tolerance of SURVEY.md 8c: PSNR >= 60 dB."""
    import evc_amd  # noqa: F401
    from evc_amd import sampler as S
    from evc_amd.config import default_config
    from evc_amd.scorenet import ScoreNet
    from oracle import samplers as OS, schedule as OSch, scorenet as ON
    cfg = default_config(32, 32, 32, subsample=100)
    d = ON.Dims(ngf=32, n_head_channels=32, image_size=32)
    p = ON.seeded_params(d, 33)
    net = ScoreNet(cfg, p)
    x_T, cond = rnd(34, 1, 15, 32, 32), rnd(35, 1, 6, 32, 32).clamp(-1, 1)
    noises = [rnd(2000 + i, 1, 15, 32, 32) for i in range(100)]
    out = S.ddpm_sampler(x_T.cuda(), net, cond=cond.cuda(), subsample_steps=100, denoise=True, clip_before=True,
                         final_only=True, noise_fn=lambda i, x: noises[i])[0].cpu()
    ref = OS.ddpm(x_T.clone(), lambda x, t: ON.forward(p, d, x, t, cond=cond), OSch.base_schedule(),
                  subsample_steps=100, noise_fn=lambda i, x: noises[i])[0]
    a, b = ((out + 1) / 2).clamp(0, 1).double(), ((ref + 1) / 2).clamp(0, 1).double()
    psnr = 10 * torch.log10(1.0 / ((a - b) ** 2).mean())
    assert float(psnr) >= 60.0, float(psnr)


def test_clip_decoder_with_the_spade_network_vs_oracle():
    """The receiver loop with the SPADE-conditioned network (``model.spade: true``): the conditioning frames change
    from chunk to chunk, so the per-chunk gamma / beta maps must be rebuilt for every generation call; generated frames
    against the oracle sampler + oracle SPADE network on the same decoded conditioning frames and injected noise.
    Weights come from the synthetic checkpoint generator (SPADE parameter names and order of the reference)."""
    import evc_amd  # noqa: F401
    from evc_amd import sampler as S, synthetic
    from evc_amd.config import default_config
    from evc_amd.decoder import ClipDecoder
    from evc_amd.elic import ElicModel
    from evc_amd.scorenet import build_score_network
    from evc_amd.scorenet_spade import SpadeScoreNet
    from oracle import samplers as OS, schedule as OSch, scorenet as ON, scorenet_spade as OSP

    cfg = default_config(32, 32, 64, subsample=3)
    cfg.model.spade = True
    cfg.model.spade_dim = 32
    d_net = ON.Dims(ngf=32, n_head_channels=32, image_size=64)
    p = synthetic.diffusion_state_dict(cfg, 19)
    assert [k for k, _ in OSP.param_shapes(d_net, spade_dim=32)] == list(p)      # reference state-dict order
    net = build_score_network(cfg, p)
    assert isinstance(net, SpadeScoreNet)
    elic = ElicModel(synthetic.elic_state_dict(4))
    dec = ClipDecoder(net, elic, cfg, S.get_sampler("DDPM"))
    B, F = 2, 12
    clips = torch.from_numpy(synthetic.make_clips(B, seed=6, frames=F, size=64).astype(np.float32) / 255)
    mask = np.array([1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0])
    key_strings, shape = [], None
    for f in (0, 1):
        enc = elic.compress(clips[:, f].cuda())
        key_strings.append(enc["strings"])
        shape = enc["shape"]
    noise_log = []

    def noise_fn(tag, shp):
        tns = rnd(2000 + len(noise_log), *shp)
        noise_log.append((tag, tns))
        return tns.cuda()
    out = dec.decode(mask, key_strings, shape, frames=F, noise_fn=noise_fn).cpu()
    inits = [i for i, (tag, _) in enumerate(noise_log) if tag == "init"]
    assert len(inits) == 2                                   # frames 2-6 and 7-11, each conditioned on the two before
    sched = OSch.base_schedule()

    def oracle_chunk(prev2, log):
        cond = 2 * prev2.reshape(B, 6, 64, 64) - 1
        steps = {tag: tns for tag, tns in log[1:]}
        x = OS.ddpm(log[0][1].clone(), lambda xx, tt: OSP.forward(p, d_net, xx, tt, cond, spade_dim=32), sched,
                    subsample_steps=3, noise_fn=lambda i, xx: steps[i])
        return ((x[0] + 1) / 2).clamp(0, 1).reshape(B, 5, 3, 64, 64)
    g1 = oracle_chunk(out[:, 0:2], noise_log[inits[0]:inits[1]])
    assert float((out[:, 2:7] - g1).abs().max()) < 2e-3
    g2 = oracle_chunk(out[:, 5:7], noise_log[inits[1]:])
    assert float((out[:, 7:12] - g2).abs().max()) < 2e-3


def test_full_size_clip_decoder_two_chunks_at_a_rank_batch_vs_oracle():
    """BASELINE configs[2]'s per-rank shape through the receiver loop at FULL size: six 128x128 clips (one rank of the
    46-clip shard), two ELIC key frames + two generated chunks (the second conditioned on frames the first produced),
    262 M-parameter network at B = 6 per launch, 2 DDPM steps + the denoise call per chunk; the first chunk is
    regenerated by the CPU oracle sampler + oracle network from the decoded key frames and the same injected noise."""
    import evc_amd  # noqa: F401
    from evc_amd import sampler as S, synthetic
    from evc_amd.config import default_config
    from evc_amd.decoder import ClipDecoder
    from evc_amd.elic import ElicModel
    from evc_amd.scorenet import ScoreNet
    from oracle import samplers as OS, schedule as OSch, scorenet as ON
    torch.set_num_threads(16)
    cfg = default_config(192, 192, 128, subsample=2)
    d_net = ON.Dims()
    p = ON.seeded_params(d_net, 1234)
    net = ScoreNet(cfg, p)
    elic = ElicModel(synthetic.elic_state_dict(3))
    dec = ClipDecoder(net, elic, cfg, S.get_sampler("DDPM"))
    B, F = 6, 12
    clips = torch.from_numpy(synthetic.make_clips(B, seed=8, frames=2).astype(np.float32) / 255)
    key_strings, shape = [], None
    for f in (0, 1):
        enc = elic.compress(clips[:, f].cuda())
        key_strings.append(enc["strings"])
        shape = enc["shape"]
    mask = np.array([1, 1] + [0] * 10)
    noise_log = []

    def noise_fn(tag, shp):
        tns = rnd(3000 + len(noise_log), *shp)
        noise_log.append((tag, tns))
        return tns.cuda()
    out = dec.decode(mask, key_strings, shape, frames=F, noise_fn=noise_fn).cpu()
    assert out.shape == (B, F, 3, 128, 128) and bool(torch.isfinite(out).all())
    from evc_amd import lib as L
    assert L.range_events() == 0          # no fp16-split operand could leave its range, no tensor held a NaN / inf
    inits = [i for i, (tag, _) in enumerate(noise_log) if tag == "init"]
    assert len(inits) == 2
    log = noise_log[inits[0]:inits[1]]
    cond = 2 * out[:, 0:2].reshape(B, 6, 128, 128) - 1
    steps = {tag: tns for tag, tns in log[1:]}
    x = OS.ddpm(log[0][1].clone(), lambda xx, tt: ON.forward(p, d_net, xx, tt, cond=cond), OSch.base_schedule(),
                subsample_steps=2, noise_fn=lambda i, xx: steps[i])
    g1 = ((x[0] + 1) / 2).clamp(0, 1).reshape(B, 5, 3, 128, 128)
    assert float((out[:, 2:7] - g1).abs().max()) < 2e-3              # fp32 sampler tolerance on [0,1] pixels
    # the second chunk was conditioned on generated frames 5, 6 (not on the key frames): it differs from the first
    assert float((out[:, 7:12] - out[:, 2:7]).abs().max()) > 1e-3


def test_clip_decoder_with_the_pseudo3d_network_vs_oracle():
    """The receiver loop with ``model.arch: unetmorepseudo3d`` (built through ``build_score_network`` from the synthetic
    checkpoint generator, whose pseudo-3-D state-dict layout is the pinned oracle's): two key frames + two generated chunks,
    the second conditioned on frames the first produced, against the oracle sampler + oracle pseudo-3-D network on the same
    decoded conditioning frames and injected noise."""
    import evc_amd  # noqa: F401
    from evc_amd import sampler as S, synthetic
    from evc_amd.config import default_config
    from evc_amd.decoder import ClipDecoder
    from evc_amd.elic import ElicModel
    from evc_amd.scorenet import build_score_network
    from evc_amd.scorenet_pseudo3d import Pseudo3dScoreNet
    from oracle import samplers as OS, schedule as OSch, scorenet_pseudo3d as O3

    cfg = default_config(32, 32, 64, subsample=3)
    cfg.model.arch, cfg.model.ch_mult, cfg.model.num_res_blocks, cfg.model.attn_resolutions = "unetmorepseudo3d", [1, 2], 1, [32]
    d_net = O3.Dims(ngf=32, ch_mult=[1, 2], num_res_blocks=1, attn_resolutions=[32], n_head_channels=32, image_size=64)
    p = synthetic.diffusion_state_dict(cfg, 23)
    assert [k for k, _ in O3.param_shapes(d_net)] == list(p)
    net = build_score_network(cfg, p)
    assert isinstance(net, Pseudo3dScoreNet)
    elic = ElicModel(synthetic.elic_state_dict(4))
    dec = ClipDecoder(net, elic, cfg, S.get_sampler("DDPM"))
    B, F = 2, 12
    clips = torch.from_numpy(synthetic.make_clips(B, seed=7, frames=F, size=64).astype(np.float32) / 255)
    mask = np.array([1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0])
    key_strings, shape = [], None
    for f in (0, 1):
        enc = elic.compress(clips[:, f].cuda())
        key_strings.append(enc["strings"])
        shape = enc["shape"]
    noise_log = []

    def noise_fn(tag, shp):
        tns = rnd(3000 + len(noise_log), *shp)
        noise_log.append((tag, tns))
        return tns.cuda()
    out = dec.decode(mask, key_strings, shape, frames=F, noise_fn=noise_fn).cpu()
    inits = [i for i, (tag, _) in enumerate(noise_log) if tag == "init"]
    assert len(inits) == 2
    sched = OSch.base_schedule()

    def oracle_chunk(prev2, log):
        cond = 2 * prev2.reshape(B, 6, 64, 64) - 1
        steps = {tag: tns for tag, tns in log[1:]}
        x = OS.ddpm(log[0][1].clone(), lambda xx, tt: O3.forward(p, d_net, xx, tt, cond=cond), sched,
                    subsample_steps=3, noise_fn=lambda i, xx: steps[i])
        return ((x[0] + 1) / 2).clamp(0, 1).reshape(B, 5, 3, 64, 64)
    g1 = oracle_chunk(out[:, 0:2], noise_log[inits[0]:inits[1]])
    assert float((out[:, 2:7] - g1).abs().max()) < 2e-3
    g2 = oracle_chunk(out[:, 5:7], noise_log[inits[1]:])
    assert float((out[:, 7:12] - g2).abs().max()) < 2e-3
