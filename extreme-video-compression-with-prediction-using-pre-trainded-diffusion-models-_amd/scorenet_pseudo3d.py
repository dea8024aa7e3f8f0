"""Pseudo-3-D score network (``config.model.arch: unetmorepseudo3d``) on MI355X.

Host-side mirror of the reference's ``NCSNpp`` with ``is3d`` / ``pseudo3d`` (models/better/ncsnpp_more.py:40-51,101-122,
213-231,259-262,328-356,387-390) and its layers (models/better/layers3d.py: ``PseudoConv3d`` :257-310, ``AttnBlockpp3d``
:191-223 with ``AttnBlockpp1d`` :81-123; models/better/layerspp.py:486-549 ``get_act_norm`` with ``is3d``, :553-624
``ResnetBlockBigGANppGN`` with the pseudo-3-D convolutions): same constructor input, same ``state_dict`` key names, same call
signature and buffers as ``ScoreNet``, so it plugs into the same samplers and decoder.

The reference keeps a video activation as (B, C*N, H, W) and reshapes inside every layer.  Here it is x[b][n][h][w][c] -- an
NHWC tensor of B*N images with a sample's N frames adjacent -- which turns every layer into a call of an existing 2-D kernel
on a VIEW of the same memory (no transposes anywhere):

* per-frame Conv2d (``space_conv``)      -> ``evc_conv2d_nhwc_f32`` over B*N images, the 3-D AdaGN + SiLU fused into its load;
* Conv1d over the frames (``time_conv``) -> ``evc_conv2d_nhwc_f32`` with a 3x1 (or 1x1) filter over B images of N rows x H*W
  columns, the SiLU between the two convolutions fused into its load, bias + residual + 1/sqrt(2) into its epilogue;
* GroupNorm over (C/G, N, H, W)          -> the same per-channel moments, taken over N*H*W pixels per sample (the time
  convolution's epilogue produces them), + ``evc_gn_coeffs_f32``; the (B, C) coefficients are repeated per frame;
* space attention                        -> ScoreNet's attention block over B*N images;
* time attention / its per-pixel GroupNorm / the N -> M frame converters -> csrc/frames.hip.

This network is not on the benchmarked path (no shipped config or checkpoint selects it): it is built
for completeness, un-fused beyond the above and on the range-free bf16x6 arithmetic; parity against the reference's own
outputs is in tests/test_gpu_scorenet.py (golden ``forward_pseudo3d.npz``).
"""
import torch

from . import lib as L
from .scorenet import FIR_K, INV_SQRT2, ScoreNet, _pad16, num_groups


def build_program_3d(d):
    """Module records in the order of NCSNpp.__init__ with is3d (ncsnpp_more.py:70-247).  Channel counts are PER FRAME (the
    reference's are these times the frame count: N = num_frames + num_frames_cond going down, M = num_frames going up)."""
    N, M = d.num_frames + d.num_frames_cond, d.num_frames
    mods = [dict(kind="linear"), dict(kind="linear")]
    res = [d.image_size // (2 ** i) for i in range(len(d.ch_mult))]
    mods.append(dict(kind="conv_in", cin=d.channels, cout=d.ngf, frames=N))
    hs_c = [d.ngf]
    in_ch = d.ngf
    for lvl, mult in enumerate(d.ch_mult):
        for _ in range(d.num_res_blocks):
            mods.append(dict(kind="res", cin=in_ch, cout=d.ngf * mult, up=False, down=False, frames=N))
            in_ch = d.ngf * mult
            if res[lvl] in d.attn_resolutions:
                mods.append(dict(kind="attn", ch=in_ch, frames=N))
            hs_c.append(in_ch)
        if lvl != len(d.ch_mult) - 1:
            mods.append(dict(kind="res", cin=in_ch, cout=in_ch, up=False, down=True, frames=N))
            hs_c.append(in_ch)
    in_ch = hs_c[-1]
    mods.append(dict(kind="res", cin=in_ch, cout=in_ch, up=False, down=False, frames=N))
    mods.append(dict(kind="attn", ch=in_ch, frames=N))
    mods.append(dict(kind="mix", ch=in_ch, frames=N, frames_out=M))               # the "converter" (:213-216)
    mods.append(dict(kind="res", cin=in_ch, cout=in_ch, up=False, down=False, frames=M))
    for lvl in reversed(range(len(d.ch_mult))):
        for _ in range(d.num_res_blocks + 1):
            skip = hs_c.pop()
            mods.append(dict(kind="mix", ch=skip, frames=N, frames_out=M))         # skip tensor N -> M frames (:226-228)
            mods.append(dict(kind="res", cin=in_ch + skip, cout=d.ngf * d.ch_mult[lvl], up=False, down=False, frames=M))
            in_ch = d.ngf * d.ch_mult[lvl]
        if res[lvl] in d.attn_resolutions:
            mods.append(dict(kind="attn", ch=in_ch, frames=M))
        if lvl != 0:
            mods.append(dict(kind="res", cin=in_ch, cout=in_ch, up=True, down=False, frames=M))
    assert not hs_c
    mods.append(dict(kind="norm", ch=in_ch, frames=M))
    mods.append(dict(kind="conv_out", cin=in_ch, cout=d.channels, frames=M))
    return mods


class _Video:
    """A video activation (B*N, H, W, C), a sample's N frames adjacent, with lazily computed per-channel moments over all
    N*H*W pixels of a sample (what the 3-D GroupNorm needs)."""
    __slots__ = ("t", "N", "_stats")

    def __init__(self, t, N, stats=None):
        self.t, self.N, self._stats = t, N, stats

    def stats(self):
        if self._stats is None:
            BN, H, W, C = self.t.shape
            self._stats = L.chan_stats(self.t.view(BN // self.N, self.N, H * W, C))
        return self._stats


class Pseudo3dScoreNet(ScoreNet):
    """HIP implementation of ``UNetMore_DDPM`` with ``arch: unetmorepseudo3d`` (eval mode, dropout 0)."""

    ARCH = "unetmorepseudo3d"

    def __init__(self, config, state_dict, device="cuda", prefix="", use_graphs=False, **_):
        super().__init__(config, state_dict, device=device, prefix=prefix, preactivate=False, use_graphs=use_graphs)
        if self.cond_emb or self.noise_in_cond:
            raise NotImplementedError("cond_emb / noise_in_cond are built for the 2-D concat-conditioned network only")
        # bf16x6 throughout (no operand-range analysis for this network): none of ScoreNet's fp16-split plumbing applies
        self._f16_raw = False
        self.fuse_skip = False
        self.overlap_skip = False

    def _build_program(self):
        return build_program_3d(self.d)

    def _embed_dim(self):
        return self.d.ngf * (self.d.num_frames + self.d.num_frames_cond)           # nf = ngf * n_frames (ncsnpp_more.py:50)

    def _pack_conv(self, w, pad_ci=None, bounded=False):
        return super()._pack_conv(w, pad_ci, bounded=False)

    # ------------------------------------------------------------------------------------------
    def _load_pconv(self, e, key, n, g, pad_ci=None, pad_mid=None):
        """PseudoConv3d parameters: ``space_conv`` (Co, Ci, k, k) and ``time_conv`` (Co, Co, k) -> a k x 1 filter."""
        ws, wt = g(n + ".space_conv.weight"), g(n + ".time_conv.weight")
        e[key + "s"] = self._pack_conv(ws, pad_ci)
        e[key + "sb"] = self._dev(g(n + ".space_conv.bias"))
        e[key + "t"] = self._pack_conv(wt[:, :, :, None], pad_mid)
        e[key + "tb"] = self._dev(g(n + ".time_conv.bias"))

    def _load(self, sd, pre):
        g = lambda name: sd[name]
        self.w = {}
        dense_w, dense_b = [], []
        off = 0
        for i, m in enumerate(self.program):
            n = pre + str(i)
            k = m["kind"]
            e = dict()
            if k == "linear":
                w = g(n + ".weight")
                e = dict(w=self._pack_conv(w[:, :, None, None]), b=self._dev(g(n + ".bias")), co=w.shape[0])
            elif k == "conv_in":
                self._load_pconv(e, "c", n, g, pad_ci=_pad16(m["cin"]))
                e["cin_pad"] = _pad16(m["cin"])
            elif k == "conv_out":     # the space convolution's 3 output channels live in a zeroed 16-channel buffer
                self._load_pconv(e, "c", n, g, pad_mid=_pad16(m["cout"]))
            elif k == "res":
                for j, key in ((0, "actnorm0"), (1, "actnorm1")):
                    dw, db = g(f"{n}.{key}.Dense_0.weight"), g(f"{n}.{key}.Dense_0.bias")
                    dense_w.append(dw); dense_b.append(db)
                    e[f"ss{j}"] = (off, dw.shape[0] // 2)
                    off += dw.shape[0]
                self._load_pconv(e, "c0", n + ".Conv_0", g)
                self._load_pconv(e, "c1", n + ".Conv_1", g)
                if m["cin"] != m["cout"] or m["up"] or m["down"]:
                    self._load_pconv(e, "c2", n + ".Conv_2", g)
            elif k == "attn":
                for part, px in (("space_att", ""), ("time_att", "t_")):
                    ws = [g(f"{n}.{part}.NIN_{j}.W") for j in range(4)]      # (in, out): conv weight is the transpose
                    bs = [g(f"{n}.{part}.NIN_{j}.b") for j in range(4)]
                    e[px + "gamma"] = self._dev(g(f"{n}.{part}.GroupNorm_0.weight"))
                    e[px + "beta"] = self._dev(g(f"{n}.{part}.GroupNorm_0.bias"))
                    e[px + "wqkv"] = self._pack_conv(torch.cat([w.t() for w in ws[:3]], 0)[:, :, None, None])
                    e[px + "bqkv"] = self._dev(torch.cat(bs[:3], 0))
                    e[px + "wo"] = self._pack_conv(ws[3].t()[:, :, None, None])
                    e[px + "bo"] = self._dev(bs[3])
            elif k == "mix":
                e = dict(w=self._dev(g(n + ".weight")[:, :, 0, 0]), b=self._dev(g(n + ".bias")))
                assert tuple(e["w"].shape) == (m["frames_out"], m["frames"])
            elif k == "norm":
                e = dict(gamma=self._dev(g(n + ".Norm_0.weight")), beta=self._dev(g(n + ".Norm_0.bias")))
            self.w[i] = e
        self.ss_total = off
        self.temb_dim = dense_w[0].shape[1]
        self.dense_w = self._pack_conv(torch.cat(dense_w, 0)[:, :, None, None])
        self.dense_b = self._dev(torch.cat(dense_b, 0))

    # ------------------------------------------------------------------------------------------
    def _pconv(self, e, key, src, co, k, N, src1=None, coef=None, act_in=L.ACT_NONE, res=None, out_scale=1.0):
        """PseudoConv3d.forward (layers3d.py:280-299): Conv2d per frame -> SiLU -> Conv1d over the frames, as two launches of
        the 2-D convolution on two views of the same memory.  Returns a ``_Video`` with its 3-D moments."""
        BN, H, W = src.shape[0], src.shape[1], src.shape[2]
        mid = None
        if co % 16:
            mid = torch.zeros((BN, H, W, _pad16(co)), device=self.device, dtype=torch.float32)
        t = L.conv2d_nhwc(src, e[key + "s"], co, k, k, bias=e[key + "sb"], src1=src1, coef=coef, act_in=act_in, out=mid)
        B, Cm = BN // N, t.shape[3]
        tv = t.view(B, N, H * W, Cm)                     # B "images" of N rows x H*W columns: the frame axis is the row axis
        rv = None if res is None else res.view(B, N, H * W, res.shape[3])
        if co % 16 == 0:
            y, st = L.conv2d_nhwc(tv, e[key + "t"], co, k, 1, bias=e[key + "tb"], act_in=L.ACT_SILU, res=rv,
                                  out_scale=out_scale, want_stats=True)
        else:
            assert rv is None
            y = L.conv2d_nhwc(tv, e[key + "t"], co, k, 1, bias=e[key + "tb"], act_in=L.ACT_SILU, out_scale=out_scale,
                              out=torch.zeros((B, N, H * W, Cm), device=self.device, dtype=torch.float32))
            st = None
        return _Video(y.view(BN, H, W, y.shape[3]), N, st)

    def _coef3d(self, coef, N):
        """(B, C) GroupNorm coefficients of a sample -> one row per frame, in the frame-major order of the activations."""
        return coef[0].repeat_interleave(N, 0), coef[1].repeat_interleave(N, 0)

    def _res3d(self, i, m, x, skip, rows):
        """ResnetBlockBigGANppGN.forward (layerspp.py:595-624) with 3-D act-norms and pseudo-3-D convolutions."""
        e, N = self.w[i], m["frames"]
        BN, H, W, _ = x.t.shape
        parts = [x.stats()] + ([skip.stats()] if skip is not None else [])
        coef0 = self._coef3d(self._adagn(parts, N * H * W, m["cin"], e["ss0"], rows), N)
        s1 = None if skip is None else skip.t
        xs, xs1 = x.t, s1
        if m["up"] or m["down"]:
            fir = (FIR_K * 4.0, 2, 1, (2, 1)) if m["up"] else (FIR_K, 1, 2, (1, 1))
            hf = L.upfirdn2d_nhwc(x.t, *fir, coef=coef0, act=L.ACT_SILU)
            xs = L.upfirdn2d_nhwc(x.t, *fir)
            h = self._pconv(e, "c0", hf, m["cout"], 3, N)
        else:
            h = self._pconv(e, "c0", x.t, m["cout"], 3, N, src1=s1, coef=coef0, act_in=L.ACT_SILU)
        H1, W1 = h.t.shape[1], h.t.shape[2]
        coef1 = self._coef3d(self._adagn([h.stats()], N * H1 * W1, m["cout"], e["ss1"], rows), N)
        if "c2s" in e:
            xs = self._pconv(e, "c2", xs, m["cout"], 1, N, src1=xs1).t
        return self._pconv(e, "c1", h.t, m["cout"], 3, N, coef=coef1, act_in=L.ACT_SILU, res=xs, out_scale=INV_SQRT2)

    def _attn3d(self, i, m, x):
        """AttnBlockpp3d.forward (layers3d.py:205-223), act = None: attention over the pixels of each frame (ScoreNet's
        block over B*N images), then over the frames of each pixel (AttnBlockpp1d, :105-123)."""
        from .scorenet import _Act
        e, N, C = self.w[i], m["frames"], m["ch"]
        t = self._attn(i, m, _Act(x.t)).t
        hd = self.d.n_head_channels
        heads = 1 if C < hd else C // hd
        y = L.frame_group_norm(t, N, e["t_gamma"], e["t_beta"], num_groups(C), 1e-6)
        qkv = L.conv2d_nhwc(y, e["t_wqkv"], 3 * C, 1, 1, bias=e["t_bqkv"])
        o = L.frame_attention(qkv, N, C, heads)
        return _Video(L.conv2d_nhwc(o, e["t_wo"], C, 1, 1, bias=e["t_bo"], res=t, out_scale=INV_SQRT2), N)

    def _mix(self, i, m, x):
        return _Video(L.frame_mix(x.t, m["frames"], self.w[i]["w"], self.w[i]["b"]), m["frames_out"])

    @torch.no_grad()
    def forward_rows(self, x, rows, cond=None):
        """x: (B, 3*num_frames, H, W) NCHW, cond: (B, 3*num_frames_cond, H, W), rows: AdaGN-table rows (B,)."""
        d, prog = self.d, self.program
        B, _, H, W = x.shape
        N, M = d.num_frames + d.num_frames_cond, d.num_frames
        taps = getattr(self, "taps", None)       # test hook: module outputs in the reference's (B, C*N, H, W) form

        def tap(idx, v):
            if taps is not None:
                c = prog[idx].get("cout", prog[idx].get("ch"))
                t5 = v.t[..., :c].reshape(B, v.N, v.t.shape[1], v.t.shape[2], c).permute(0, 4, 1, 2, 3)
                taps[idx] = t5.reshape(-1, v.N, *t5.shape[3:]) if prog[idx]["kind"] == "mix" else \
                    t5.reshape(B, c * v.N, *t5.shape[3:])
            return v
        it = iter(range(2, len(prog)))

        def run(h, skip=None):
            i = next(it)
            m = prog[i]
            if m["kind"] == "mix":                   # the skip tensor's converter precedes its res-block (:344-351)
                skip = tap(i, self._mix(i, m, skip))
                i = next(it)
                m = prog[i]
            if m["kind"] == "res":
                return tap(i, self._res3d(i, m, h, skip, rows))
            if m["kind"] == "attn":
                return tap(i, self._attn3d(i, m, h))
            raise AssertionError(m)
        # (B, N*3, H, W) -> B*N images of 3 channels, frames adjacent (the reference's B, N*C -> B, C*N permute, :259-262)
        frames = x if cond is None else torch.cat([x, cond], dim=1)
        assert frames.shape[1] == N * d.channels, (tuple(frames.shape), N)
        i = next(it)
        xin = L.pack_nchw_to_nhwc(frames.reshape(B * N, d.channels, H, W), None, self.w[i]["cin_pad"])
        hs = [tap(i, self._pconv(self.w[i], "c", xin, prog[i]["cout"], 3, N))]
        n_lvl = len(d.ch_mult)
        for lvl in range(n_lvl):
            for _ in range(d.num_res_blocks):
                h = run(hs[-1])
                if h.t.shape[2] in d.attn_resolutions:
                    h = run(h)
                hs.append(h)
            if lvl != n_lvl - 1:
                hs.append(run(hs[-1]))
        h = run(hs[-1])
        h = run(h)
        i = next(it)
        h = tap(i, self._mix(i, prog[i], h))
        h = run(h)
        for lvl in reversed(range(n_lvl)):
            for _ in range(d.num_res_blocks + 1):
                h = run(h, hs.pop())
            if h.t.shape[2] in d.attn_resolutions:
                h = run(h)
            if lvl != 0:
                h = run(h)
        assert not hs
        # final 3-D GroupNorm + SiLU fused into the output convolution's load (ncsnpp_more.py:380-388)
        i = next(it)
        e = self.w[i]
        Hh, Wh, C = h.t.shape[1], h.t.shape[2], h.t.shape[3]
        coef = self._coef3d(L.gn_coeffs([h.stats()], M * Hh * Wh, num_groups(C), 1e-5, mode=1, gamma=e["gamma"],
                                        beta=e["beta"]), M)
        i = next(it)
        co = prog[i]["cout"]
        out = self._pconv(self.w[i], "c", h.t, co, 3, M, coef=coef, act_in=L.ACT_SILU)
        assert next(it, None) is None
        res = L.nhwc_to_nchw(out.t, co).reshape(B, M * co, Hh, Wh)           # frames major: the reference's :387-390 permute
        if taps is not None:
            taps[i] = out.t[..., :co].reshape(B, M, Hh, Wh, co).permute(0, 4, 1, 2, 3).reshape(B, co * M, Hh, Wh)
        return res


class Conv3dScoreNet(Pseudo3dScoreNet):
    """``arch: unetmore3d``: the same network with ``MyConv3d`` (nn.Conv3d 3x3x3 / 1x1x1, layers3d.py:225-254) where the
    pseudo-3-D variant has its Conv2d -> SiLU -> Conv1d pairs (ncsnpp_more.py:104-106, layerspp.py:569-571).

    A 3x3x3 convolution over (N, H, W) is ONE 3x3 convolution over B*N images whose input channels are the activated frames
    n - 1 | n | n + 1 side by side (``evc_frame_taps_f32``: zeros beyond a sample's first / last frame -- which is why the
    act-norm is materialised first instead of fused into the load: a missing frame must stay zero), with the weight
    (Co, Ci, kt, kh, kw) read as (Co, kt*Ci + ci, kh, kw).  The 1x1x1 convolution is a 1x1 convolution over B*N images."""

    ARCH = "unetmore3d"

    def _load_pconv(self, e, key, n, g, pad_ci=None, pad_mid=None):
        w = g(n + ".conv.weight")                                  # (Co, Ci, k, k, k)
        co, ci, k = w.shape[0], w.shape[1], w.shape[2]
        cp = ci if pad_ci is None else pad_ci
        w2 = torch.zeros((co, k, cp, k, k), dtype=torch.float32)
        w2[:, :, :ci] = w.detach().float().permute(0, 2, 1, 3, 4)
        e[key + "s"] = self._pack_conv(w2.reshape(co, k * cp, k, k))
        e[key + "sb"] = self._dev(g(n + ".conv.bias"))

    def _pconv(self, e, key, src, co, k, N, src1=None, coef=None, act_in=L.ACT_NONE, res=None, out_scale=1.0):
        BN, H, W, C0 = src.shape
        if k == 3:
            C1 = 0 if src1 is None else src1.shape[3]
            if coef is not None or src1 is not None:
                act = torch.empty((BN, H, W, C0 + C1), device=self.device, dtype=torch.float32)
                L.affine_act(src, coef, act_in, out=L.Cols(act, 0, C0))
                if src1 is not None:
                    L.affine_act(src1, coef, act_in, out=L.Cols(act, C0, C1), coef_col=C0)
                src = act
            src, src1, coef, act_in = L.frame_taps(src, N), None, None, L.ACT_NONE
        out = torch.zeros((BN, H, W, _pad16(co)), device=self.device, dtype=torch.float32) if co % 16 else None
        r = L.conv2d_nhwc(src, e[key + "s"], co, k, k, bias=e[key + "sb"], src1=src1, coef=coef, act_in=act_in, res=res,
                          out_scale=out_scale, out=out, want_stats=(out is None))
        if out is not None:
            return _Video(r, N)
        y, st = r
        return _Video(y, N, st.view(BN // N, N * st.shape[1], st.shape[2], 2))    # per-image moments = N times the pixel runs
