"""Score network (NCSN++ "unetmore") on MI355X.

Host-side mirror of the reference's ``UNetMore_DDPM`` / ``NCSNpp`` (models/better/ncsnpp_more.py:32-392,
721-770): same constructor input (a config namespace), same ``state_dict`` key names
(``unet.all_modules.<i>...``), same call signature ``net(x, labels, cond=None, cond_mask=None) -> eps`` with
NCHW float32 tensors, same ``alphas / alphas_prev / betas`` buffers the samplers read
(models/__init__.py:223).  Everything between the NCHW boundary tensors runs in NHWC through the HIP
kernels of libevc_hip.so:

* 3x3 / 1x1 convolutions, NIN and the Linear layers -> ``evc_conv2d_nhwc_f32`` (implicit GEMM on the f32
  matrix cores) with the preceding GroupNorm affine + SiLU fused into its operand load, the skip
  concat read in place (two sources), and bias + residual + 1/sqrt(2) fused into its epilogue;
* GroupNorm -> per-channel moments (``evc_chan_stats_f32``, computed once per tensor and reused for every
  grouping it takes part in) + ``evc_gn_coeffs_f32``;
* FIR up/down sampling -> ``evc_upfirdn2d_nhwc_f32`` with the AdaGN + SiLU of the h-branch fused on load;
* q/k/v NINs as ONE 1x1 conv (Co = 3C) feeding ``evc_attention_f32``;
* the time-embedding MLP and all 70 AdaGN ``Dense_0`` projections depend only on the label, so they are
  evaluated once per distinct label into a table row -- 72 GEMVs per forward become 0.
"""
import math
from types import SimpleNamespace

import numpy as np
import torch

from . import lib as L

FIR_K = np.outer([1, 3, 3, 1], [1, 3, 3, 1]).astype(np.float32)
FIR_K /= FIR_K.sum()
INV_SQRT2 = float(np.float32(1.0) / np.sqrt(np.float32(2.0)))


def num_groups(ch):
    """reference models/better/layerspp.py:473-476."""
    g = min(ch // 4, 32)
    while ch % g != 0:
        g -= 1
    return g


def dims_from_config(config):
    m, d = config.model, config.data
    return SimpleNamespace(ngf=m.ngf, ch_mult=list(m.ch_mult), num_res_blocks=m.num_res_blocks,
                           attn_resolutions=list(m.attn_resolutions), n_head_channels=m.n_head_channels,
                           image_size=d.image_size, channels=d.channels, num_frames=d.num_frames,
                           num_frames_cond=d.num_frames_cond + getattr(d, "num_frames_future", 0),
                           sigma_begin=getattr(m, "sigma_begin", 0.02), sigma_end=getattr(m, "sigma_end", 1e-4),
                           num_classes=getattr(m, "num_classes", 1000))


def build_program(d):
    """Module records in the order of NCSNpp.__init__ (ncsnpp_more.py:70-247)."""
    mods = [dict(kind="linear"), dict(kind="linear")]
    res = [d.image_size // (2 ** i) for i in range(len(d.ch_mult))]
    mods.append(dict(kind="conv_in", cin=d.channels * (d.num_frames + d.num_frames_cond), cout=d.ngf))
    hs_c = [d.ngf]
    in_ch = d.ngf
    for lvl, mult in enumerate(d.ch_mult):
        for _ in range(d.num_res_blocks):
            mods.append(dict(kind="res", cin=in_ch, cout=d.ngf * mult, up=False, down=False))
            in_ch = d.ngf * mult
            if res[lvl] in d.attn_resolutions:
                mods.append(dict(kind="attn", ch=in_ch))
            hs_c.append(in_ch)
        if lvl != len(d.ch_mult) - 1:
            mods.append(dict(kind="res", cin=in_ch, cout=in_ch, up=False, down=True))
            hs_c.append(in_ch)
    in_ch = hs_c[-1]
    mods.append(dict(kind="res", cin=in_ch, cout=in_ch, up=False, down=False))
    mods.append(dict(kind="attn", ch=in_ch))
    mods.append(dict(kind="res", cin=in_ch, cout=in_ch, up=False, down=False))
    for lvl in reversed(range(len(d.ch_mult))):
        for _ in range(d.num_res_blocks + 1):
            skip = hs_c.pop()
            mods.append(dict(kind="res", cin=in_ch + skip, cout=d.ngf * d.ch_mult[lvl], up=False, down=False,
                             split=(in_ch, skip)))
            in_ch = d.ngf * d.ch_mult[lvl]
        if res[lvl] in d.attn_resolutions:
            mods.append(dict(kind="attn", ch=in_ch))
        if lvl != 0:
            mods.append(dict(kind="res", cin=in_ch, cout=in_ch, up=True, down=False))
    assert not hs_c
    mods.append(dict(kind="norm", ch=in_ch))
    mods.append(dict(kind="conv_out", cin=in_ch, cout=d.channels * d.num_frames))
    return mods


class _Act:
    """An NHWC activation with lazily computed, cached per-channel moments."""
    __slots__ = ("t", "_stats")

    def __init__(self, t, stats=None):
        self.t = t
        self._stats = stats

    def stats(self):
        if self._stats is None:
            self._stats = L.chan_stats(self.t)
        return self._stats


def _pad16(c):
    return (c + 15) // 16 * 16


class DdpmWrapper:
    """What ``UNetMore_DDPM.__init__`` / ``forward`` (ncsnpp_more.py:735-768) and ``UNet_DDPM`` (models/unet.py:337-372) do
    around the network itself, shared by ScoreNet and UNetDDPM: the schedule buffers the samplers index (linear or cosine),
    the Gamma-noise buffers, and ``noise_in_cond``."""

    def _init_wrapper(self, m, sigma_begin, sigma_end, num_classes):
        self.noise_in_cond = bool(getattr(m, "noise_in_cond", False))
        self.cond_noise_fn = None        # tests inject the draw of noise_in_cond here: fn(cond) -> raw noise tensor
        self.cond_generator = None
        # schedule buffers exactly as the reference builds them (CPU float32)
        dist = getattr(m, "sigma_dist", "linear")
        if dist == "linear":
            self.betas = torch.linspace(sigma_begin, sigma_end, num_classes)
            self.alphas = torch.cumprod(1 - self.betas.flip(0), 0).flip(0)
            self.alphas_prev = torch.cat([self.alphas[1:], torch.tensor([1.0]).to(self.alphas)])
        elif dist == "cosine":           # models/__init__.py:29-33 (get_sigmas) are the alphas, betas derived
            T = num_classes
            t = torch.linspace(T, 0, T + 1) / T
            f = torch.cos((t + 0.008) / (1 + 0.008) * math.pi / 2) ** 2
            self.alphas = f[:-1] / f[-1]
            self.alphas_prev = torch.cat([self.alphas[1:], torch.tensor([1.0]).to(self.alphas)])
            self.betas = 1 - self.alphas / self.alphas_prev
        else:
            raise NotImplementedError(f"sigma_dist {dist!r}: the reference builds its buffers for 'linear' and 'cosine' only")
        self.gamma = bool(getattr(m, "gamma", False))
        if self.gamma:       # Gamma-noise schedule buffers (read by the samplers' gamma=True branch)
            self.theta_0 = 0.001
            self.k = self.betas / (self.alphas * (self.theta_0 ** 2))
            self.k_cum = torch.cumsum(self.k.flip(0), 0).flip(0)
            self.theta_t = torch.sqrt(self.alphas) * self.theta_0

    def _noised_cond(self, cond, vals):
        """noise_in_cond: cond <- sqrt(a) cond + sqrt(1 - a) z with a = alphas[label] per sample, z Gaussian -- or, on a
        gamma model, a standardised Gamma(k_cum[label], rate 1/theta_t[label]) draw -- fresh on every call, as the reference
        draws it in every forward.  ``cond_noise_fn`` (tests) supplies the raw draw."""
        if not self.noise_in_cond or cond is None:
            return cond
        if any(v != int(v) or v < 0 for v in vals):
            raise IndexError("noise_in_cond indexes the schedule with the labels (alphas[labels]): integer labels only")
        raw = None if self.cond_noise_fn is None else self.cond_noise_fn(cond).to(self.device, torch.float32).contiguous()
        out = torch.empty_like(cond)
        start = 0
        while start < len(vals):                     # runs of equal labels (the samplers: one run)
            end = start
            while end < len(vals) and vals[end] == vals[start]:
                end += 1
            i = int(vals[start])
            a = self.alphas[i]
            c = cond[start:end].contiguous()
            if self.gamma:
                from .sampler import _gamma_noise
                z = _gamma_noise(c, self.k_cum[i], self.theta_t[i], a, raw=None if raw is None else raw[start:end],
                                 generator=self.cond_generator)
            elif raw is not None:
                z = raw[start:end].contiguous()
            else:
                z = torch.randn(c.shape, device=self.device, dtype=torch.float32, generator=self.cond_generator)
            out[start:end] = L.lincomb4([c, z], [float(a.sqrt()), float((1 - a).sqrt())])
            start = end
        return out


class ScoreNet(DdpmWrapper):
    """HIP implementation of ``UNetMore_DDPM`` (eval mode, dropout 0), incl. its cond_emb / noise_in_cond / gamma / cosine-schedule
    options (ncsnpp_more.py:61,97-99,282-285,735-768)."""

    SPADE = False      # scorenet_spade.SpadeScoreNet: conditioning through SPADE act-norms (model.spade: true)
    ARCH = "unetmore"  # scorenet_pseudo3d.Pseudo3dScoreNet: "unetmorepseudo3d"

    def __init__(self, config, state_dict, device="cuda", prefix="", preactivate=False, use_graphs=False):
        L.hip_lib()   # fail loudly before touching anything else
        # preactivate=False: AdaGN + SiLU is fused into the 3x3 convolutions' operand load (evaluated once per
        # filter tap, costs MFMA issue slots); True: applied once per tensor by evc_affine_act_nhwc_f32 and the
        # convolutions read the activated tensor as is (one extra HBM round trip per tensor).  Same values either
        # way; on MI355X at B=9 the two tie (38.6 vs 38.4 ms per forward, interleaved A/B), so the fused form
        # with less memory traffic is the default.
        self.preactivate = preactivate
        self.config = config
        self.device = torch.device(device)
        self.d = dims_from_config(config)
        self.type = getattr(config.model, "type", "v1")
        m = config.model
        if bool(getattr(m, "spade", False)) != self.SPADE or getattr(m, "output_all_frames", False) or m.arch != self.ARCH:
            raise NotImplementedError("built: arch=unetmore -- the concat-conditioned network of configs/mine.yml (ScoreNet) and its "
                                      "SPADE variant (SpadeScoreNet) -- and arch=unetmorepseudo3d / unetmore3d without SPADE "
                                      "(Pseudo3dScoreNet / Conv3dScoreNet), all without output_all_frames (which fails in the "
                                      "reference itself: ncsnpp_more.py:384-385 splits 15 channels into 6 + 15)")
        # cond_emb: the time embedding is extended by an Embedding(2, ngf // 2) row chosen by cond_mask (ncsnpp_more.py:97-99,
        # :282-285); noise_in_cond, the schedule (linear / cosine) and the Gamma buffers: DdpmWrapper
        self.cond_emb = bool(getattr(m, "cond_emb", False))
        self._init_wrapper(m, self.d.sigma_begin, self.d.sigma_end, self.d.num_classes)
        if (self.cond_emb or self.noise_in_cond) and self.SPADE:
            raise NotImplementedError("cond_emb / noise_in_cond are built for the concat-conditioned network only")
        self.embed_dim = self._embed_dim()
        self.program = self._build_program()
        if self.SPADE:     # the conditioning frames do not enter through the input (ncsnpp_more.py:519, :593-594)
            self.program[2]["cin"] = self.d.channels * self.d.num_frames
        self._load(state_dict, prefix + "unet.all_modules.")
        self._rows = {}          # label value -> row of the AdaGN table
        self._row_tensors = {}   # (row, B) -> int32 device tensor
        # AdaGN table with spare capacity: captured graphs hold pointers into it, so it only moves (and the
        # graphs are dropped) when more than `capacity` distinct labels have been seen.
        self._table = torch.zeros((256, self.ss_total), device=self.device, dtype=torch.float32)
        self._n_rows = 0
        # Optional: one captured HIP graph per (stream, batch, height, width, has-cond).  Replay cuts the host
        # cost of a forward (~450 launches) from 4.8 ms to 0.2 ms, bit-identical results -- but on MI355X the
        # forward is GPU-bound at every batch size measured (B=1: 16.2 ms, B=9: 38.6 ms, same with and without),
        # so it is off by default; useful when the host thread has other work.
        self.use_graphs = use_graphs
        self._graphs = {}
        # fp16-split convolutions on raw (un-normalised) inputs take an element bound from the tensors' moments
        self._f16_raw = L.bounded_arith() == L.ARITH_F16X3
        # the x-branch of a res-block (1x1 skip convolution) on a side stream, overlapped with the h-branch
        import os
        self.overlap_skip = os.environ.get("EVC_OVERLAP_SKIP", "1") != "0"
        # a res-block's 1x1 skip convolution (Conv_2) as extra K-steps of its Conv_1 launch (one accumulator, no separate
        # launch / output / residual re-read): f16x3 row-reuse kernel only, EVC_FUSE_SKIP=0 restores the separate launches
        self.fuse_skip = self._f16_raw and os.environ.get("EVC_FUSE_SKIP", "1") != "0"
        self._side_streams = {}
        self._bounds_by_stream = {}      # concurrent clip groups run forwards on their own streams: one arena each
        self._bounds = None
        self._bound_next = 0

    # ------------------------------------------------------------------------------------------
    def _build_program(self):
        return build_program(self.d)

    def _embed_dim(self):
        """Width of the sinusoidal time embedding (``nf``, ncsnpp_more.py:50,80,274)."""
        return self.d.ngf

    def _dev(self, t):
        return t.detach().to(device=self.device, dtype=torch.float32).contiguous()

    def _pack_conv(self, w, pad_ci=None, bounded=False):
        """``bounded``: the convolution's input is GroupNorm-normalised / activated (O(1)), so the fp16-split
        arithmetic applies; raw residual-stream inputs keep the bf16 split (no range assumption)."""
        w = self._dev(w)
        if pad_ci is not None and pad_ci != w.shape[1]:
            wp = torch.zeros((w.shape[0], pad_ci, w.shape[2], w.shape[3]), device=self.device)
            wp[:, :w.shape[1]] = w
            w = wp
        return L.conv_pack_weights(w, L.bounded_arith() if bounded else L.default_arith())

    def _load(self, sd, pre):
        g = lambda name: sd[name]
        self.w = {}
        dense_w, dense_b = [], []
        off = 0
        if self.cond_emb:                # modules[2] = Embedding(2, ngf // 2): every later state-dict index is one higher
            self.cond_table = self._dev(g(pre + "2.weight"))
            assert tuple(self.cond_table.shape) == (2, self.d.ngf // 2), tuple(self.cond_table.shape)
        for i, m in enumerate(self.program):
            n = pre + str(i + (1 if self.cond_emb and i >= 2 else 0))
            k = m["kind"]
            if k == "linear":
                w = g(n + ".weight")
                self.w[i] = dict(w=self._pack_conv(w[:, :, None, None]), b=self._dev(g(n + ".bias")),
                                 co=w.shape[0])
            elif k in ("conv_in", "conv_out"):
                w = g(n + ".weight")     # conv_out reads the final GroupNorm + SiLU; conv_in the raw network input
                self.w[i] = dict(w=self._pack_conv(w, _pad16(w.shape[1]), bounded=(k == "conv_out")),
                                 b=self._dev(g(n + ".bias")),
                                 co=w.shape[0], cin_pad=_pad16(w.shape[1]))
            elif k == "res":
                e = dict()
                for j, key in ((0, "actnorm0"), (1, "actnorm1")):
                    dw, db = g(f"{n}.{key}.Dense_0.weight"), g(f"{n}.{key}.Dense_0.bias")
                    dense_w.append(dw); dense_b.append(db)
                    e[f"ss{j}"] = (off, dw.shape[0] // 2)
                    off += dw.shape[0]
                    self._load_actnorm(e, j, f"{n}.{key}", g)
                # Conv_0 / Conv_1 read AdaGN + SiLU outputs (directly or through the FIR resampler): O(1) operands
                e["w0"] = self._pack_conv(g(n + ".Conv_0.weight"), bounded=True); e["b0"] = self._dev(g(n + ".Conv_0.bias"))
                e["w1"] = self._pack_conv(g(n + ".Conv_1.weight"), bounded=True); e["b1"] = self._dev(g(n + ".Conv_1.bias"))
                if m["cin"] != m["cout"] or m["up"] or m["down"]:
                    # 1x1 skip convolution on the raw residual stream: fp16 split too, scaled by the element bound that
                    # the block's own GroupNorm moments give (gn_coeffs(..., bound=))
                    e["w2"] = self._pack_conv(g(n + ".Conv_2.weight"), bounded=True); e["b2"] = self._dev(g(n + ".Conv_2.bias"))
                    e["b12"] = (e["b1"] + e["b2"]).contiguous()      # bias of Conv_1 with Conv_2 fused into it
                self.w[i] = e
            elif k == "attn":
                ws = [g(f"{n}.NIN_{j}.W") for j in range(4)]   # (in, out): conv weight is the transpose
                bs = [g(f"{n}.NIN_{j}.b") for j in range(4)]
                wqkv = torch.cat([w.t() for w in ws[:3]], 0)[:, :, None, None]
                self.w[i] = dict(gamma=self._dev(g(n + ".GroupNorm_0.weight")),
                                 beta=self._dev(g(n + ".GroupNorm_0.bias")),
                                 wqkv=self._pack_conv(wqkv, bounded=True),     # input: affine GroupNorm
                                 bqkv=self._dev(torch.cat(bs[:3], 0)),
                                 # output projection: |attention output| <= max |v|, bounded through v's moments
                                 wo=self._pack_conv(ws[3].t()[:, :, None, None], bounded=True), bo=self._dev(bs[3]))
            elif k == "norm":
                self.w[i] = self._load_final_norm(n, g)
        self.ss_total = off
        self.temb_dim = dense_w[0].shape[1]
        self.dense_w = self._pack_conv(torch.cat(dense_w, 0)[:, :, None, None])
        self.dense_b = self._dev(torch.cat(dense_b, 0))

    def _load_actnorm(self, e, j, name, g):
        """Hook: extra parameters of act-norm ``j`` of a res-block (none here; SPADE maps in SpadeScoreNet)."""

    def _load_final_norm(self, n, g):
        return dict(gamma=self._dev(g(n + ".Norm_0.weight")), beta=self._dev(g(n + ".Norm_0.bias")))

    # ------------------------------------------------------------------------------------------
    def _embedding(self, labels):
        """get_timestep_embedding (models/better/layers.py:504-518) on the host, float32."""
        dim = self.embed_dim
        half = dim // 2
        emb = math.log(10000) / (half - 1)
        emb = torch.exp(torch.arange(half, dtype=torch.float32) * -emb)
        emb = torch.tensor(labels, dtype=torch.float32)[:, None] * emb[None, :]
        return torch.cat([torch.sin(emb), torch.cos(emb)], dim=1)

    def _key(self, label, mask=1):
        """Key of an AdaGN-table row: the label, and with cond_emb the conditioning flag that picks the embedding row."""
        return (float(label), int(mask)) if self.cond_emb else float(label)

    def prepare_labels(self, labels, masks=None):
        """Evaluate temb MLP + every AdaGN Dense_0 for the labels not in the table yet (one batched pass).  ``masks``: with
        cond_emb, the cond_mask value of each label (default 1, as the reference does for cond_mask=None)."""
        masks = [1] * len(labels) if masks is None else masks
        new = [k for k in dict.fromkeys(self._key(x, mk) for x, mk in zip(labels, masks)) if k not in self._rows]
        if not new:
            return
        R = len(new)
        e = self._embedding([k[0] if self.cond_emb else k for k in new]).to(self.device).reshape(1, 1, R, self.embed_dim).contiguous()
        w0, w1 = self.w[0], self.w[1]
        t = L.conv2d_nhwc(e, w0["w"], w0["co"], 1, 1, bias=w0["b"])                       # Linear(ngf -> 4ngf)
        t = L.conv2d_nhwc(t, w1["w"], w1["co"], 1, 1, bias=w1["b"], act_in=L.ACT_SILU)    # act -> Linear
        if self.cond_emb:                # temb = cat([temb, Embedding(cond_mask)]) (ncsnpp_more.py:282-285); rows are label-major
            emb = self.cond_table[torch.tensor([k[1] for k in new], device=self.device)]
            t = torch.cat([t.reshape(R, -1), emb], 1).reshape(1, 1, R, -1).contiguous()
        rows = L.conv2d_nhwc(t, self.dense_w, self.ss_total, 1, 1, bias=self.dense_b, act_in=L.ACT_SILU)
        base = self._n_rows
        if base + R > self._table.shape[0]:
            grown = torch.zeros((max(2 * self._table.shape[0], base + R), self.ss_total), device=self.device)
            grown[:base] = self._table[:base]
            self._table = grown
            self._graphs.clear()        # captured graphs point into the old table
        self._table[base:base + R] = rows.reshape(R, self.ss_total)
        self._n_rows = base + R
        for j, v in enumerate(new):
            self._rows[v] = base + j

    def _row_tensor(self, row_ids):
        key = tuple(row_ids)
        t = self._row_tensors.get(key)
        if t is None:
            t = torch.tensor(row_ids, dtype=torch.int32, device=self.device)
            self._row_tensors[key] = t
        return t

    # ------------------------------------------------------------------------------------------
    def _adagn(self, parts, hw, ch, seg, rows, bound=None):
        off, c = seg
        assert c == ch
        return L.gn_coeffs(parts, hw, num_groups(ch), 1e-5, mode=2, ss=self._table[:, off:off + 2 * c], row=rows,
                           bound=bound)

    def _side_stream(self, main):
        """The side stream paired with a main stream (one per concurrent clip group / capture)."""
        key = main.cuda_stream
        st = self._side_streams.get(key)
        if st is None:
            st = self._side_streams[key] = torch.cuda.Stream(device=self.device)
        return st

    def _bound_slot(self, n=1):
        """n zeroed device words of this forward's arena (element bounds for the fp16-split kernels on raw inputs)."""
        if not self._f16_raw:
            return None
        w = self._bounds[self._bound_next:self._bound_next + n]
        self._bound_next += n
        return w

    def _res(self, i, m, x, skip, rows):
        """ResnetBlockBigGANppGN.forward (models/better/layerspp.py:595-624)."""
        e = self.w[i]
        B, H, W, _ = x.t.shape
        parts = [x.stats()] + ([skip.stats()] if skip is not None else [])
        xbound = self._bound_slot() if "w2" in e else None     # bounds x (and skip): also FIR(x), whose taps sum to 1
        coef0 = self._adagn(parts, H * W, m["cin"], e["ss0"], rows, bound=xbound)
        fir = None
        if m["up"] or m["down"]:
            fir = (FIR_K * 4.0, 2, 1, (2, 1)) if m["up"] else (FIR_K, 1, 2, (1, 1))   # upsample_2d: gain factor**2, pad (2, 1)
                                                                                      # downsample_2d: pad (1, 1)
        s1 = None if skip is None else skip.t

        def skip_path(out):
            """x-branch of the block: [FIR ->] 1x1 skip convolution (models/better/layerspp.py:603-614)."""
            src, src1 = x.t, s1
            if fir is not None:
                src, src1 = L.upfirdn2d_nhwc(x.t, *fir), None
            if "w2" in e:
                return L.conv2d_nhwc(src, e["w2"], m["cout"], 1, 1, bias=e["b2"], src1=src1, in_bound=xbound, out=out)
            return src

        Ho, Wo = (2 * H, 2 * W) if m["up"] else ((H // 2, W // 2) if m["down"] else (H, W))
        fuse = self.fuse_skip and "w2" in e and not self.preactivate and \
            L.conv_fused_1x1_supported(B, Ho, Wo, m["cout"], m["cout"], L.packed_arith(e["w1"]))
        # The x-branch only depends on the block input: with `overlap_skip` it runs on a side stream while the main stream
        # does the h-branch (FIR, Conv_0, its moments), so its small latency-bound launches (the 1x1 convolution, its
        # split-K combine) hide behind Conv_0 instead of sitting in front of Conv_1.  Joined by an event before Conv_1.
        # With `fuse` only the FIR of an up / down block is left of it (the 1x1 convolution rides in Conv_1's launch).
        side = done = xs = None
        if self.overlap_skip and "w2" in e and (not fuse or fir is not None):
            main = torch.cuda.current_stream()
            side = self._side_stream(main)
            xs = torch.empty((B, Ho, Wo, m["cin"] if fuse else m["cout"]), device=self.device, dtype=torch.float32)   # owned by the main stream
            side.wait_stream(main)                # coef0's launch raised the bound word; x / skip are complete
            with torch.cuda.stream(side):
                if fuse:
                    L.upfirdn2d_nhwc(x.t, *fir, out=xs)
                else:
                    skip_path(xs)
                done = torch.cuda.Event()
                done.record(side)
        if fir is not None:
            hf = L.upfirdn2d_nhwc(x.t, *fir, coef=coef0, act=L.ACT_SILU)
            h1 = _Act(*L.conv2d_nhwc(hf, e["w0"], m["cout"], 3, 3, bias=e["b0"], want_stats=True))
        elif self.preactivate:
            cin = m["cin"]
            act = torch.empty((B, H, W, cin), device=self.device, dtype=torch.float32)
            c0 = x.t.shape[3]
            L.affine_act(x.t, coef0, L.ACT_SILU, out=L.Cols(act, 0, c0))
            if skip is not None:
                L.affine_act(skip.t, coef0, L.ACT_SILU, out=L.Cols(act, c0, cin - c0), coef_col=c0)
            h1 = _Act(*L.conv2d_nhwc(act, e["w0"], m["cout"], 3, 3, bias=e["b0"], want_stats=True))
        else:
            h1 = _Act(*L.conv2d_nhwc(x.t, e["w0"], m["cout"], 3, 3, bias=e["b0"], src1=s1, coef=coef0, act_in=L.ACT_SILU,
                                     want_stats=True))
        H1, W1 = h1.t.shape[1], h1.t.shape[2]
        coef1 = self._adagn([h1.stats()], H1 * W1, m["cout"], e["ss1"], rows)
        if side is None:
            if fuse:
                xs = L.upfirdn2d_nhwc(x.t, *fir) if fir is not None else None
            else:
                xs = skip_path(None)
        else:
            torch.cuda.current_stream().wait_event(done)
        if fuse:
            # Conv_1 + Conv_2 in one launch (models/better/layerspp.py:603-624): x2 = the block input (both halves of a skip
            # concat read in place) or its FIR-resampled copy; |FIR(x)| <= max |x| (taps sum to 1), so x's bound serves both
            x2 = (xs, None, e["w2"], xbound) if xs is not None else (x.t, s1, e["w2"], xbound)
            return _Act(*L.conv2d_nhwc(h1.t, e["w1"], m["cout"], 3, 3, bias=e["b12"], coef=coef1, act_in=L.ACT_SILU,
                                       out_scale=INV_SQRT2, want_stats=True, x2=x2))
        if self.preactivate:
            h1a = L.affine_act(h1.t, coef1, L.ACT_SILU)
            return _Act(*L.conv2d_nhwc(h1a, e["w1"], m["cout"], 3, 3, bias=e["b1"], res=xs, out_scale=INV_SQRT2,
                                       want_stats=True))
        return _Act(*L.conv2d_nhwc(h1.t, e["w1"], m["cout"], 3, 3, bias=e["b1"], coef=coef1, act_in=L.ACT_SILU,
                                   res=xs, out_scale=INV_SQRT2, want_stats=True))

    def _attn(self, i, m, x):
        """AttnBlockpp.forward (models/better/layerspp.py:230-249)."""
        e = self.w[i]
        B, H, W, C = x.t.shape
        hd = self.d.n_head_channels
        heads = 1 if C < hd else C // hd
        coef = L.gn_coeffs([x.stats()], H * W, num_groups(C), 1e-6, mode=1, gamma=e["gamma"], beta=e["beta"])
        qkvb = self._bound_slot(3)
        if qkvb is None:
            obound = None
            qkv = L.conv2d_nhwc(x.t, e["wqkv"], 3 * C, 1, 1, bias=e["bqkv"], coef=coef)
        else:
            # element bounds of q, k, v from the projection's fused moments (one small launch); the attention output is
            # a convex combination of value rows, so max |o| <= max |v|: v's bound also serves the output projection
            qkv, qst = L.conv2d_nhwc(x.t, e["wqkv"], 3 * C, 1, 1, bias=e["bqkv"], coef=coef, want_stats=True)
            L.moments_bound(qst, 0, C, qkvb)
            obound = qkvb[2:3]
        o = L.attention(qkv.view(B, H * W, 3 * C), C, heads, bounds=qkvb)
        return _Act(*L.conv2d_nhwc(o.view(B, H, W, C), e["wo"], C, 1, 1, bias=e["bo"], res=x.t,
                                   out_scale=INV_SQRT2, want_stats=True, in_bound=obound))

    @torch.no_grad()
    def forward_rows(self, x, rows, cond=None):
        """x: (B, C*num_frames, H, W) NCHW, rows: int32 (B,) device tensor of AdaGN-table rows."""
        d = self.d
        prog = self.program
        B, _, H, W = x.shape
        if self._f16_raw:
            # one arena per stream (concurrent clip groups run their forwards on their own streams).  Inside a HIP-graph
            # capture every graph owns its arena: all captures share one capture stream, and graphs replayed
            # concurrently on different streams would race on a shared one.
            if torch.cuda.is_current_stream_capturing():
                self._bounds = torch.empty(256, device=self.device, dtype=torch.int32)
            else:
                key = torch.cuda.current_stream().cuda_stream
                self._bounds = self._bounds_by_stream.get(key)
                if self._bounds is None:
                    self._bounds = self._bounds_by_stream[key] = torch.empty(256, device=self.device, dtype=torch.int32)
            self._bounds.zero_()          # one memset per forward; slots are handed out in program order
            self._bound_next = 0
        # test hook (tests/test_gpu_scorenet.py): ``self.taps = {}`` before a call collects every module's output, NCHW,
        # under its index in the reference's ``all_modules`` list (ncsnpp_more.py:251-392)
        taps = getattr(self, "taps", None)

        def tap(idx, act):
            if taps is not None:
                taps[idx] = act.t[..., :prog[idx].get("cout", prog[idx].get("ch"))].permute(0, 3, 1, 2).contiguous()
            return act
        i = 2
        m = prog[i]
        xin = self._pack_input(x, cond, self.w[i]["cin_pad"])
        hs = [tap(i, _Act(*L.conv2d_nhwc(xin, self.w[i]["w"], m["cout"], 3, 3, bias=self.w[i]["b"], want_stats=True)))]
        i += 1
        n_lvl = len(d.ch_mult)
        for lvl in range(n_lvl):
            for _ in range(d.num_res_blocks):
                h = tap(i, self._res(i, prog[i], hs[-1], None, rows)); i += 1
                if h.t.shape[2] in d.attn_resolutions:
                    h = tap(i, self._attn(i, prog[i], h)); i += 1
                hs.append(h)
            if lvl != n_lvl - 1:
                hs.append(tap(i, self._res(i, prog[i], hs[-1], None, rows))); i += 1
        h = hs[-1]
        h = tap(i, self._res(i, prog[i], h, None, rows)); i += 1
        h = tap(i, self._attn(i, prog[i], h)); i += 1
        h = tap(i, self._res(i, prog[i], h, None, rows)); i += 1
        for lvl in reversed(range(n_lvl)):
            for _ in range(d.num_res_blocks + 1):
                h = tap(i, self._res(i, prog[i], h, hs.pop(), rows)); i += 1
            if h.t.shape[2] in d.attn_resolutions:
                h = tap(i, self._attn(i, prog[i], h)); i += 1
            if lvl != 0:
                h = tap(i, self._res(i, prog[i], h, None, rows)); i += 1
        assert not hs
        out, co = self._final(i, h)
        assert i + 2 == len(prog)
        res = L.nhwc_to_nchw(out, co)
        if taps is not None:
            taps[i + 1] = res
        return res

    def _pack_input(self, x, cond, cin_pad):
        """Network input: frames to denoise and conditioning frames concatenated along channels (ncsnpp_more.py:256-257)."""
        return L.pack_nchw_to_nhwc(x, cond, cin_pad)

    def _final(self, i, h):
        """Final GroupNorm + SiLU fused into the output convolution's load (ncsnpp_more.py:380-388)."""
        e = self.w[i]
        B, H, W, C = h.t.shape
        coef = L.gn_coeffs([h.stats()], H * W, num_groups(C), 1e-5, mode=1, gamma=e["gamma"], beta=e["beta"])
        co = self.program[i + 1]["cout"]
        out = torch.empty((B, H, W, _pad16(co)), device=h.t.device, dtype=torch.float32)
        L.conv2d_nhwc(h.t, self.w[i + 1]["w"], co, 3, 3, bias=self.w[i + 1]["b"], coef=coef, act_in=L.ACT_SILU, out=out)
        return out, co

    def _forward(self, x, rows, cond):
        """``forward_rows`` through a captured HIP graph when enabled (static input buffers, one replay)."""
        if not self.use_graphs:
            return self.forward_rows(x, rows, cond)
        # static buffers belong to one stream: concurrent clip groups on different streams get their own graph
        key = (torch.cuda.current_stream().cuda_stream, tuple(x.shape), cond is not None, self.preactivate)
        g = self._graphs.get(key)
        if g is None:
            g = self._capture(x, rows, cond)
            self._graphs[key] = g
        g["x"].copy_(x)
        g["rows"].copy_(rows)
        if cond is not None:
            g["cond"].copy_(cond)
        g["graph"].replay()
        return g["out"].clone()     # the static output buffer is overwritten by the next replay

    def _capture(self, x, rows, cond):
        sx, srows = x.clone(), rows.clone()
        scond = None if cond is None else cond.clone()
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):     # warm-up outside capture: workspaces, allocator pools
            self.forward_rows(sx, srows, scond)
        cur.wait_stream(side)
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = self.forward_rows(sx, srows, scond)
        return dict(graph=graph, x=sx, rows=srows, cond=scond, out=out)

    def forward_label(self, x, label, cond=None):
        """All samples share one label (what every sampler does): no device->host sync."""
        self.prepare_labels([label])
        rows = self._row_tensor([self._rows[self._key(label)]] * x.shape[0])
        if self.noise_in_cond and cond is not None:
            cond = self._noised_cond(cond.to(self.device, torch.float32).contiguous(), [float(label)] * x.shape[0])
        return self._forward(x, rows, cond)

    def __call__(self, x, labels, cond=None, cond_mask=None):
        """Reference call shape: ``scorenet(x, labels, cond=cond)`` (models/__init__.py:265,285); ``cond_mask`` (one flag per
        sample, default ones) only matters with cond_emb."""
        vals = [float(v) for v in (labels.detach().cpu().tolist() if torch.is_tensor(labels) else labels)]
        assert len(vals) == x.shape[0]
        masks = None
        if self.cond_emb and cond_mask is not None:
            masks = [int(v) for v in (cond_mask.detach().cpu().tolist() if torch.is_tensor(cond_mask) else cond_mask)]
        self.prepare_labels(vals, masks)
        rows = self._row_tensor([self._rows[self._key(v, 1 if masks is None else masks[j])] for j, v in enumerate(vals)])
        x = x.to(self.device, torch.float32).contiguous()
        cond = None if cond is None else cond.to(self.device, torch.float32).contiguous()
        return self._forward(x, rows, self._noised_cond(cond, vals))

    forward = __call__

    def eval(self):
        return self


def build_score_network(config, state_dict, device="cuda", **kw):
    """``config.model.arch``: "unetmore" -> ScoreNet (the network the reference CLI hard-codes, city_sender.py:311-312), or
    SpadeScoreNet when ``config.model.spade`` (ncsnpp_more.py:730-733); "unet" -> UNetDDPM (reference models/unet.py,
    upstream MCVD's name for it); "unetmorepseudo3d" / "unetmore3d" -> Pseudo3dScoreNet / Conv3dScoreNet (the is3d / pseudo3d
    branches of ncsnpp_more.py with models/better/layers3d.py).  All plug into the same samplers."""
    arch = getattr(config.model, "arch", "unetmore")
    if arch == "unet":
        from .unet_ddpm import UNetDDPM
        return UNetDDPM(config, state_dict, device=device)
    if arch in ("unetmorepseudo3d", "unetmore3d"):
        from .scorenet_pseudo3d import Conv3dScoreNet, Pseudo3dScoreNet
        return (Conv3dScoreNet if arch == "unetmore3d" else Pseudo3dScoreNet)(config, state_dict, device=device, **kw)
    if getattr(config.model, "spade", False):
        from .scorenet_spade import SpadeScoreNet
        return SpadeScoreNet(config, state_dict, device=device, **kw)
    return ScoreNet(config, state_dict, device=device, **kw)
