"""SPADE-conditioned NCSN++ score network (``model.spade: true``) on the HIP kernels.

Reference: ``models/better/ncsnpp_more.py:396-718`` (``SPADE_NCSNpp``), ``models/better/layerspp.py:101-173``
(``MySPADE``), ``:486-549`` (``get_act_norm`` with ``norm == 'spade'``), ``:628-705`` (``ResnetBlockBigGANppSPADE``);
selected by ``UNetMore_DDPM.__init__`` (``ncsnpp_more.py:730-733``).  SURVEY.md section 8 row f4 (alt model; the
shipped ``configs/mine.yml:117`` has ``spade: false``).

What differs from ``ScoreNet``: the conditioning frames do not enter through the input concat; every act-norm is

    SiLU( [ GroupNorm_noaffine(x, eps 1e-6) * (1 + gamma(cond)) + beta(cond) ] * (1 + scale(t)) + shift(t) )

with per-pixel ``gamma / beta = conv3x3(SiLU(conv3x3(nearest_resize(cond))))``.  MI355X-first restructuring:

* the gamma / beta maps depend on the conditioning frames only, not on x or on the step label: they are computed ONCE per
  chunk (first forward that sees a new ``cond`` tensor) and reused by the other 100 forwards of the chunk -- 3 of the 5
  convolutions of every act-norm leave the step loop (the reference recomputes them every step).  ``1 + gamma`` and
  ``beta`` come out of ONE convolution with 2C output channels (the ``1`` folded into its bias);
* per-pixel scale / shift cannot ride on the convolutions' per-channel operand-load transform, so the act-norm is one
  elementwise pass (``evc_spade_act_nhwc_f32``: GroupNorm coefficients from the producers' fused moments, maps, AdaGN
  row, SiLU) and the 3x3 convolutions read the activated tensor as is.

Parity: ``tests/test_gpu_scorenet.py::test_spade_*`` against ``tests/golden/forward_spade.npz`` (imported reference).
"""
import torch

from . import lib as L
from .scorenet import FIR_K, INV_SQRT2, ScoreNet, _Act, _pad16, num_groups


class SpadeScoreNet(ScoreNet):
    SPADE = True

    def __init__(self, config, state_dict, device="cuda", prefix="", **kw):
        kw.pop("preactivate", None)
        kw["use_graphs"] = False           # the per-chunk map cache is filled inside the first forward
        self.spade_dim = getattr(config.model, "spade_dim", 128)
        if self.spade_dim % 16:
            raise NotImplementedError("spade_dim must be a multiple of 16 (convolution K-step)")
        self._cond_ref = None              # the cond tensor the cached maps were computed from (kept alive: identity test)
        self._cond_version = -1
        self._maps = {}
        self._segs = {}
        super().__init__(config, state_dict, device=device, prefix=prefix, **kw)
        self.overlap_skip = False

    # ---- parameters ----------------------------------------------------------------------------
    def _pack_conv(self, w, pad_ci=None, bounded=False):
        """Every 3x3 convolution of this network (Conv_0, Conv_1, the output convolution, the map convolutions) reads a
        MATERIALISED tensor -- the SPADE act-norm output, norm(x) (1 + gamma_map) + beta_map then scale / shift / SiLU, or
        resized conditioning frames -- whose range no coefficient kernel has bounded (gn_coeffs runs in mode 0 here and
        sees only the unmodulated norm(x); evc_spade_act raises no range event).  The fp16 split's "range_events == 0 proves
        every operand in range" therefore does not cover them: they stay on the exact bf16 split, which assumes no range.
        The 1x1 convolutions (skip path with a moments bound, attention projections behind an affine GroupNorm) keep it."""
        if w.shape[-1] == 3:
            bounded = False
        return super()._pack_conv(w, pad_ci, bounded)

    def _spade_params(self, name, g):
        """mlp_shared / mlp_gamma / mlp_beta of one MySPADE (layerspp.py:147-150): gamma and beta share their input, so
        their filters are stacked into one convolution whose first half carries the ``1 +`` in its bias."""
        ws = g(name + ".mlp_shared.0.weight")
        wg, wb = g(name + ".mlp_gamma.weight"), g(name + ".mlp_beta.weight")
        bg, bb = self._dev(g(name + ".mlp_gamma.bias")), self._dev(g(name + ".mlp_beta.bias"))
        assert ws.shape[0] == self.spade_dim and wg.shape[1] == self.spade_dim
        return dict(ws=self._pack_conv(ws, _pad16(ws.shape[1])), bs=self._dev(g(name + ".mlp_shared.0.bias")),
                    cpad=_pad16(ws.shape[1]), wgb=self._pack_conv(torch.cat([wg, wb], 0)),
                    bgb=torch.cat([bg + 1.0, bb], 0).contiguous(), ch=wg.shape[0])

    def _load_actnorm(self, e, j, name, g):
        e[f"sp{j}"] = self._spade_params(name + ".Norm_0", g)

    def _load_final_norm(self, n, g):
        return dict(sp0=self._spade_params(n + ".Norm_0", g))

    # ---- conditioning maps (once per chunk) ----------------------------------------------------
    def _set_cond(self, cond):
        if cond is None:
            raise ValueError("the SPADE network needs the conditioning frames (cond=)")
        if cond is self._cond_ref and cond._version == self._cond_version:
            return
        self._cond_ref, self._cond_version = cond, cond._version
        self._maps.clear()
        self._segs.clear()

    def _seg(self, H, W):
        """Conditioning frames at H x W, NHWC, channels zero-padded to the convolution's K-step:
        F.interpolate(mode='nearest') of layerspp.py:164 picks source pixel floor(i * H0 / H) = i * (H0 / H)."""
        t = self._segs.get((H, W))
        if t is None:
            cond = self._cond_ref
            B, C, H0, W0 = cond.shape
            if H0 % H or W0 % W:
                raise NotImplementedError("nearest resize by a non-integer factor")
            full = self._segs.get((H0, W0))
            if full is None:
                full = self._segs[(H0, W0)] = L.pack_nchw_to_nhwc(cond, None, _pad16(C))
            t = full if (H, W) == (H0, W0) else full[:, ::H0 // H, ::W0 // W, :].contiguous()
            self._segs[(H, W)] = t
        return t

    def _map(self, key, sp, B, H, W):
        """[1 + gamma | beta] of one act-norm at its resolution, (B, H, W, 2C)."""
        m = self._maps.get(key)
        if m is None:
            seg = self._seg(H, W)
            actv = L.conv2d_nhwc(seg, sp["ws"], self.spade_dim, 3, 3, bias=sp["bs"], act_out=L.ACT_SILU)
            m = self._maps[key] = L.conv2d_nhwc(actv, sp["wgb"], 2 * sp["ch"], 3, 3, bias=sp["bgb"])
        return m

    # ---- forward -------------------------------------------------------------------------------
    def _pack_input(self, x, cond, cin_pad):
        self._set_cond(cond)
        return L.pack_nchw_to_nhwc(x, None, cin_pad)

    def _actnorm(self, key, sp, parts, seg, rows, bound=None):
        """get_act_norm.forward, norm == 'spade' (layerspp.py:518-549) on a (virtual) concat of NHWC parts."""
        B, H, W, _ = parts[0].t.shape
        ch = sp["ch"]
        coef = L.gn_coeffs([p.stats() for p in parts], H * W, num_groups(ch), 1e-6, mode=0, bound=bound)
        maps = self._map(key, sp, B, H, W)
        ss = None
        if seg is not None:
            off, c = seg
            assert c == ch
            ss = self._table[:, off:off + 2 * c]
        out = torch.empty((B, H, W, ch), device=self.device, dtype=torch.float32)
        col = 0
        for p in parts:
            c = p.t.shape[3]
            L.spade_act(p.t, coef, maps, ch, col, ss=ss, row=rows if ss is not None else None, out=L.Cols(out, col, c))
            col += c
        return out

    def _res(self, i, m, x, skip, rows):
        """ResnetBlockBigGANppSPADE.forward (models/better/layerspp.py:675-705)."""
        e = self.w[i]
        parts = [x] + ([skip] if skip is not None else [])
        xbound = self._bound_slot() if "w2" in e else None
        h = self._actnorm((i, 0), e["sp0"], parts, e["ss0"], rows, bound=xbound)
        fir = None
        if m["up"] or m["down"]:
            fir = (FIR_K * 4.0, 2, 1, (2, 1)) if m["up"] else (FIR_K, 1, 2, (1, 1))
            h = L.upfirdn2d_nhwc(h, *fir)
        h1 = _Act(*L.conv2d_nhwc(h, e["w0"], m["cout"], 3, 3, bias=e["b0"], want_stats=True))
        h1a = self._actnorm((i, 1), e["sp1"], [h1], e["ss1"], rows)
        src, src1 = x.t, (None if skip is None else skip.t)
        if fir is not None:
            src, src1 = L.upfirdn2d_nhwc(x.t, *fir), None
        xs = src
        if "w2" in e:
            xs = L.conv2d_nhwc(src, e["w2"], m["cout"], 1, 1, bias=e["b2"], src1=src1, in_bound=xbound)
        return _Act(*L.conv2d_nhwc(h1a, e["w1"], m["cout"], 3, 3, bias=e["b1"], res=xs, out_scale=INV_SQRT2,
                                   want_stats=True))

    def _final(self, i, h):
        """Final SPADE act-norm (no time embedding) + output convolution (ncsnpp_more.py:700-707)."""
        act = self._actnorm((i, 0), self.w[i]["sp0"], [h], None, None)
        B, H, W, _ = act.shape
        co = self.program[i + 1]["cout"]
        out = torch.empty((B, H, W, _pad16(co)), device=self.device, dtype=torch.float32)
        L.conv2d_nhwc(act, self.w[i + 1]["w"], co, 3, 3, bias=self.w[i + 1]["b"], out=out)
        return out, co
