"""Receiver: decode a batch of 30-frame clips from key-frame bitstreams + the transmit mask ``d``.

The reference never separates sender and receiver: ``city_sender.py:521-550`` runs the generator inside the
sender and keeps only the bit count.  The receiver side implied by that loop is restated here: frames with
``d == 1`` are ELIC key frames (decoded from their strings), runs of ``d == 0`` are generated, up to 5 at a
time, by the diffusion sampler conditioned on the last two decoded frames
(``SenderCity.update`` / ``generate_frame``, city_sender.py:326-351, 408-437), always 5 frames per call, of
which the receiver keeps as many as the mask says.

Clips are independent, so they are stacked along the batch axis of every kernel launch: all clips of a batch
must share the same mask (they do in the benchmark pattern; the sender policy can group by mask).
"""
import numpy as np
import torch

from . import lib as L
from .elic import count_bits


def all_generated_mask(frames=30, key=2, chunk=5):
    """2 key frames then generated chunks (the "maximum of 5 generation cycles"+ pattern of ret/readme.md:38)."""
    d = np.zeros(frames, dtype=np.int64)
    d[:key] = 1
    return d


class ClipDecoder:
    def __init__(self, scorenet, elic_model, config, sampler, groups=1):
        self.net, self.elic, self.config, self.sampler = scorenet, elic_model, config, sampler
        self.device = scorenet.device
        self.groups = groups          # concurrent clip groups (HIP streams) during generation
        self._stream_pool = []

    @torch.no_grad()
    def generate(self, cond_frames, noise_fn=None, generator=None, groups=None):
        """cond_frames: (B, 2, 3, H, W) in [0, 1] on the device -> (B, 5, 3, H, W) in [0, 1].
        = SenderCity.generate_frame (city_sender.py:326-351) without the per-chunk checkpoint reload.

        ``groups`` > 1 splits the batch into that many clip groups that are sampled concurrently, each on its
        own HIP stream (clips are independent): idle CUs during one group's small kernels / partial tile rounds
        run another group's convolutions.  Per-clip results do not depend on the grouping when noise is injected
        (``noise_fn`` is then called per group with the group's slice bounds)."""
        from . import sampler as S
        cfg = self.config
        B, _, C, H, W = cond_frames.shape
        groups = self.groups if groups is None else groups
        groups = max(1, min(int(groups), B))
        if getattr(self.net, "SPADE", False):
            # the SPADE network caches its per-chunk gamma / beta maps for ONE conditioning tensor: interleaved clip groups
            # would each pass their own slice and rebuild all maps on every forward (~20 ms against a 15 ms forward)
            groups = 1
        cond = cond_frames.reshape(B, -1, H, W).contiguous()
        if cfg.data.rescaled:
            cond = L.scale_clamp(cond, 2.0, -1.0)                          # data_transform: 2x - 1
        ch = cfg.data.channels * cfg.data.num_frames
        kw = dict(final_only=True, denoise=cfg.sampling.denoise, subsample_steps=getattr(cfg.sampling, "subsample", None),
                  clip_before=getattr(cfg.sampling, "clip_before", True))
        step_gen = S.get_step_generator(self.sampler)

        def draw(tag, lo, hi, gen_):
            shp = (hi - lo, ch, H, W)
            if noise_fn is not None:
                return noise_fn(tag, (B, ch, H, W))[lo:hi].to(self.device).contiguous()
            return torch.randn(shp, device=self.device, dtype=torch.float32, generator=gen_)

        if groups == 1 or step_gen is None:
            x_T = draw("init", 0, B, generator)
            step_noise = None if noise_fn is None else (lambda i, x: draw(i, 0, B, None))
            out = self.sampler(x_T, self.net, cond=cond, noise_fn=step_noise, generator=generator, **kw)
            pred = out[-1].contiguous()
        else:
            bounds = [(g * B // groups, (g + 1) * B // groups) for g in range(groups)]
            main = torch.cuda.current_stream()
            streams = self._streams(groups)
            if hasattr(self.net, "prepare_labels"):
                # AdaGN table rows are shared state: build every row this sampler will read (F-PNDM: incl. the
                # Runge-Kutta midpoints and -1) on the main stream, which all group streams wait on below
                self.net.prepare_labels(S.label_set(self.sampler, self.net, kw["subsample_steps"], kw["denoise"]))
            gens = []
            for (lo, hi), st in zip(bounds, streams):
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    gen_g = None
                    if noise_fn is None:
                        gen_g = torch.Generator(device=self.device)
                        gen_g.manual_seed(int(torch.randint(0, 2 ** 31 - 1, (1,), generator=generator, device=self.device))
                                          if generator is not None else torch.seed() % (2 ** 31))
                    x_T = draw("init", lo, hi, gen_g)
                    sn = None if noise_fn is None else (lambda i, x, lo=lo, hi=hi: draw(i, lo, hi, None))
                    gens.append(step_gen(x_T, self.net, cond=cond[lo:hi].contiguous(), noise_fn=sn, generator=gen_g, **kw))
            outs = S.run_interleaved(gens, streams)
            for st in streams:
                main.wait_stream(st)
            pred = torch.cat([o[-1] for o in outs], dim=0).contiguous()
        pred = L.scale_clamp(pred, 0.5, 0.5, (0.0, 1.0)) if cfg.data.rescaled else \
            L.scale_clamp(pred, 1.0, 0.0, (0.0, 1.0))                         # inverse_data_transform
        return pred.reshape(B, cfg.data.num_frames, C, H, W)

    def _streams(self, n):
        while len(self._stream_pool) < n:
            self._stream_pool.append(torch.cuda.Stream(device=self.device))
        return self._stream_pool[:n]

    @torch.no_grad()
    def decode(self, d, key_strings, shape, frames=30, noise_fn=None, generator=None):
        """d: (frames,) 0/1 mask shared by the batch; key_strings: list over key-frame positions (in order) of
        ``[y_strings, z_strings]`` for the whole batch.  Returns (B, frames, 3, H, W) float32 on the device."""
        d = np.asarray(d).reshape(-1)
        out = []
        k = 0
        t = 0
        n_key = int(d[:frames].sum())
        assert len(key_strings) >= n_key, "not enough key-frame bitstreams for the mask"
        while t < frames:
            if d[t] == 1:
                # decode every consecutive key frame of this run in one batched ELIC call
                run = 0
                while t + run < frames and d[t + run] == 1:
                    run += 1
                ys = [[[s for f in range(run) for s in key_strings[k + f][0][i][p]] for p in range(2)]
                      for i in range(len(key_strings[k][0]))]
                zs = [s for f in range(run) for s in key_strings[k + f][1]]
                x_hat = self.elic.decompress([ys, zs], shape)["x_hat"]                  # (run*B, 3, H, W)
                B = x_hat.shape[0] // run
                x_hat = x_hat.reshape(run, B, *x_hat.shape[1:]).permute(1, 0, 2, 3, 4)
                out.append(x_hat)
                k += run
                t += run
            else:
                assert t >= 2, "a generated frame needs two decoded frames before it"
                run = 0
                while t + run < frames and d[t + run] == 0 and run < self.config.data.num_frames:
                    run += 1
                prev = torch.cat(out, dim=1)[:, -2:]
                gen = self.generate(prev.contiguous(), noise_fn=noise_fn, generator=generator)
                out.append(gen[:, :run])
                t += run
        return torch.cat(out, dim=1)[:, :frames].contiguous()


def total_bits(key_strings):
    return sum(count_bits(s) for s in key_strings)
