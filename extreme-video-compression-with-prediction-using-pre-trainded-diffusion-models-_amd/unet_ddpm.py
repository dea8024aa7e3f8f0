"""The reference's alternative score network ``UNet_DDPM`` (models/unet.py:184-371) on MI355X.

Same constructor input (a config namespace), same ``state_dict`` key names (``unet.downblocks.<i>...``), same call
``net(x, labels, cond=None) -> eps`` and the ``alphas / alphas_prev / betas`` buffers the samplers read, so it drops
into ``sampler.ddpm_sampler`` / ``ClipDecoder`` in place of ``ScoreNet`` (the shipped CLI never
builds this network -- it is the file BASELINE.json names; section 8f item 4).  Built from the same HIP kernels:

* every 3x3 convolution -> ``evc_conv2d_nhwc_f32`` with the preceding GroupNorm(32, eps 1e-6, affine) + Swish fused into
  its operand load, the up-path concat read in place, the residual (or its 1x1 ``Nin`` projection) added in the epilogue;
* the time embedding only depends on the label: ``temb_dense`` and every block's ``dense`` projection are evaluated once
  per distinct label into a table row that already contains ``conv0.bias + dense(temb)``, so the per-sample bias of
  ``h += dense(temb)`` (models/unet.py:93-94) is just the bias pointer of the convolution;
* stride-2 convolutions = "same" convolution + decimation (``evc_upfirdn2d_nhwc_f32``, 1x1 kernel, down 2);
  ``nn.Upsample(nearest)`` = ``evc_upfirdn2d_nhwc_f32`` with a 2x2 box kernel, up 2, pad (1, 0);
* attention = one head of width C (``AttnBlock``, models/unet.py:102-123): Q | K | V ``Nin`` as one 1x1 conv,
  ``evc_attention_f32`` / ``evc_attention_f16x3_f32``, ``OUT`` with the residual.  The attention kernels exist for head
  widths 32, 64, 128, 192 and 256 (``ATTENTION_WIDTHS`` below, csrc/attention.hip); this network attends at width 2*ngf
  (and at the deepest level's width in the middle block), so it runs for ``model.ngf`` in {16, 32, 64, 96, 128} with
  ``mode: deep`` -- any other width, including the mine.yml value ngf = 192 (attention at 384 channels in ONE head), raises
  ``NotImplementedError`` from the constructor, before any kernel is launched.
"""
import math

import numpy as np
import torch

from . import lib as L
from .scorenet import DdpmWrapper, _Act, _pad16

ATTENTION_WIDTHS = (32, 64, 128, 192, 256)      # head widths attention_dispatch (csrc/attention.hip) instantiates

BOX2 = np.ones((2, 2), dtype=np.float32)
ONE = np.ones((1, 1), dtype=np.float32)


def ch_mults(ngf, mode):
    return [ngf * n for n in {"deep": (1, 2, 2, 2), "deeper": (1, 2, 2, 4, 4), "deepest": (1, 2, 2, 2, 4, 4)}[mode]]


def build_program(ngf, mode, n_in):
    """Module records in the order of UNet.__init__ (models/unet.py:214-255): (list name, index, record)."""
    cm = ch_mults(ngf, mode)
    down, mid, up = [dict(kind="conv", cin=n_in, cout=ngf, stride=1)], [], []
    prev, ch_size = cm[0], [ngf]
    for i, ich in enumerate(cm):
        for first in (prev, ich):
            down.append(dict(kind="res", cin=first, cout=ich))
            ch_size.append(ich)
            if i == 1:
                down.append(dict(kind="attn", ch=ich))
        if i != len(cm) - 1:
            down.append(dict(kind="conv", cin=ich, cout=ich, stride=2))
            ch_size.append(ich)
        prev = ich
    mid += [dict(kind="res", cin=cm[-1], cout=cm[-1]), dict(kind="attn", ch=cm[-1]), dict(kind="res", cin=cm[-1], cout=cm[-1])]
    prev = cm[-1]
    for i, ich in reversed(list(enumerate(cm))):
        for _ in range(3):
            skip = ch_size.pop()
            up.append(dict(kind="res", cin=prev + skip, cout=ich, split=(prev, skip)))
            if i == 1:
                up.append(dict(kind="attn", ch=ich))
            prev = ich
        if i != 0:
            up.append(dict(kind="upsample", ch=ich))
    return [("downblocks", j, m) for j, m in enumerate(down)] + [("middleblocks", j, m) for j, m in enumerate(mid)] + \
           [("upblocks", j, m) for j, m in enumerate(up)]


class UNetDDPM(DdpmWrapper):
    """HIP implementation of ``UNet_DDPM`` (eval mode, dropout 0; output_all_frames off)."""

    def __init__(self, config, state_dict, device="cuda", prefix=""):
        m_ = config.model
        mode_ = getattr(config, "mode", "deep")
        widths = sorted({m["ch"] for _, _, m in build_program(m_.ngf, mode_, 1) if m["kind"] == "attn"})
        bad = [w for w in widths if w not in ATTENTION_WIDTHS]
        if bad:
            raise NotImplementedError(
                f"UNetDDPM with ngf={m_.ngf} (mode {mode_!r}) attends in one head of width "
                f"{bad}; the HIP attention kernels exist for widths {list(ATTENTION_WIDTHS)} only, i.e. ngf in (16, 32, 64, 96, 128) "
                f"with mode 'deep'.  (The network the reference CLI builds, arch 'unetmore', has 192-wide heads and is "
                f"fully supported.)")
        L.hip_lib()
        m, d = config.model, config.data
        if getattr(m, "output_all_frames", False):
            raise NotImplementedError("output_all_frames is not built (off in configs/mine.yml)")
        self.config, self.device = config, torch.device(device)
        self.ngf = m.ngf
        self.mode = getattr(config, "mode", "deep")
        self.time_conditional = bool(getattr(m, "time_conditional", False))
        self.affine_input = not getattr(d, "logit_transform", False) and not getattr(d, "rescaled", False)
        self.channels, self.num_frames = d.channels, d.num_frames
        n_cond = d.num_frames_cond + getattr(d, "num_frames_future", 0)
        self.n_in = d.channels * (d.num_frames + n_cond)
        self.type = getattr(m, "type", "v1")
        # schedule (linear / cosine), Gamma buffers, noise_in_cond: models/unet.py:337-372, same code as UNetMore_DDPM
        self._init_wrapper(m, getattr(m, "sigma_begin", 0.02), getattr(m, "sigma_end", 1e-4), getattr(m, "num_classes", 1000))
        self.program = build_program(self.ngf, self.mode, self.n_in)
        self._load(state_dict, prefix + "unet.")
        self._rows, self._row_bias = {}, {}
        self._f16_raw = L.bounded_arith() == L.ARITH_F16X3
        self._bounds_by_stream = {}
        self._bounds, self._bound_next = None, 0

    # ------------------------------------------------------------------------------------------
    def _dev(self, t):
        return t.detach().to(device=self.device, dtype=torch.float32).contiguous()

    def _pack(self, w, pad_ci=None, bounded=False):
        w = self._dev(w)
        if pad_ci is not None and pad_ci != w.shape[1]:
            wp = torch.zeros((w.shape[0], pad_ci, w.shape[2], w.shape[3]), device=self.device)
            wp[:, :w.shape[1]] = w
            w = wp
        return L.conv_pack_weights(w, L.bounded_arith() if bounded else L.default_arith())

    def _load(self, sd, pre):
        g = lambda n: sd[n]
        self.w = {}
        dense_w, dense_b, base_bias = [], [], []
        off = 0
        for lst, j, m in self.program:
            n, key = f"{pre}{lst}.{j}", (lst, j)
            if m["kind"] == "conv":
                w = g(n + ".weight")
                first = lst == "downblocks" and j == 0
                self.w[key] = dict(w=self._pack(w, _pad16(w.shape[1]) if first else None), b=self._dev(g(n + ".bias")),
                                   cin_pad=_pad16(w.shape[1]))
            elif m["kind"] == "upsample":
                self.w[key] = dict(w=self._pack(g(n + ".conv.weight")), b=self._dev(g(n + ".conv.bias")))
            elif m["kind"] == "res":
                e = dict(g0=self._dev(g(n + ".normalize0.weight")), be0=self._dev(g(n + ".normalize0.bias")),
                         w0=self._pack(g(n + ".conv0.weight"), bounded=True), b0=self._dev(g(n + ".conv0.bias")),
                         g1=self._dev(g(n + ".normalize1.weight")), be1=self._dev(g(n + ".normalize1.bias")),
                         w1=self._pack(g(n + ".conv1.weight"), bounded=True), b1=self._dev(g(n + ".conv1.bias")))
                if self.time_conditional:
                    dense_w.append(g(n + ".dense.weight")); dense_b.append(g(n + ".dense.bias"))
                    base_bias.append(g(n + ".conv0.bias"))
                    e["tseg"] = (off, m["cout"])
                    off += m["cout"]
                if m["cin"] != m["cout"]:
                    e["wn"] = self._pack(g(n + ".nin.weights")[:, :, None, None], bounded=True)
                    e["bn"] = self._dev(g(n + ".nin.bias"))
                self.w[key] = e
            elif m["kind"] == "attn":
                wq = torch.cat([g(f"{n}.{nm}.weights") for nm in ("Q", "K", "V")], 0)[:, :, None, None]
                self.w[key] = dict(gamma=self._dev(g(n + ".normalize.weight")), beta=self._dev(g(n + ".normalize.bias")),
                                   wqkv=self._pack(wq, bounded=True),
                                   bqkv=self._dev(torch.cat([g(f"{n}.{nm}.bias") for nm in ("Q", "K", "V")], 0)),
                                   wo=self._pack(g(n + ".OUT.weights")[:, :, None, None], bounded=True),
                                   bo=self._dev(g(n + ".OUT.bias")))
        self.norm = dict(gamma=self._dev(g(pre + "normalize.weight")), beta=self._dev(g(pre + "normalize.bias")))
        wo = g(pre + "out.weight")
        self.out = dict(w=self._pack(wo, bounded=True), b=self._dev(g(pre + "out.bias")), co=wo.shape[0])
        self.t_total = off
        if self.time_conditional:
            self.t0 = dict(w=self._pack(g(pre + "temb_dense.0.weight")[:, :, None, None]), b=self._dev(g(pre + "temb_dense.0.bias")))
            self.t1 = dict(w=self._pack(g(pre + "temb_dense.2.weight")[:, :, None, None]), b=self._dev(g(pre + "temb_dense.2.bias")))
            self.dense_w = self._pack(torch.cat(dense_w, 0)[:, :, None, None])
            # table row = conv0.bias + dense.bias + dense.weight . temb  ->  the convolution's bias for that label
            self.dense_b = self._dev(torch.cat(dense_b, 0) + torch.cat(base_bias, 0))

    # ------------------------------------------------------------------------------------------
    def _embedding(self, labels):
        half = self.ngf // 2
        emb = math.log(10000) / (half - 1)
        emb = torch.exp(torch.arange(half, dtype=torch.float32) * -emb)
        emb = torch.tensor(labels, dtype=torch.float32)[:, None] * emb[None, :]
        return torch.cat([torch.sin(emb), torch.cos(emb)], dim=1)

    def prepare_labels(self, labels):
        """temb_dense + every block's dense projection for the labels not seen yet (one batched pass per call)."""
        if not self.time_conditional:
            return
        new = [float(v) for v in dict.fromkeys(float(x) for x in labels) if float(v) not in self._row_bias]
        if not new:
            return
        R = len(new)
        e = self._embedding(new).to(self.device).reshape(1, 1, R, self.ngf).contiguous()
        t = L.conv2d_nhwc(e, self.t0["w"], 4 * self.ngf, 1, 1, bias=self.t0["b"])
        t = L.conv2d_nhwc(t, self.t1["w"], 4 * self.ngf, 1, 1, bias=self.t1["b"], act_in=L.ACT_SILU)
        rows = L.conv2d_nhwc(t, self.dense_w, self.t_total, 1, 1, bias=self.dense_b, act_in=L.ACT_SILU).reshape(R, self.t_total)
        for j, v in enumerate(new):
            self._row_bias[v] = rows[j].contiguous()

    def _bound_slot(self, n=1):
        if not self._f16_raw:
            return None
        w = self._bounds[self._bound_next:self._bound_next + n]
        self._bound_next += n
        return w

    def _gn(self, parts, hw, ch, gamma, beta, bound=None):
        return L.gn_coeffs(parts, hw, 32, 1e-6, mode=1, gamma=gamma, beta=beta, bound=bound)

    def _res(self, key, m, x, skip, trow):
        """ResnetBlock.forward (models/unet.py:88-99); ``skip``: the tensor the up path concatenates behind ``x``."""
        e = self.w[key]
        B, H, W, _ = x.t.shape
        parts = [x.stats()] + ([skip.stats()] if skip is not None else [])
        xb = self._bound_slot() if "wn" in e else None
        coef0 = self._gn(parts, H * W, m["cin"], e["g0"], e["be0"], bound=xb)
        s1 = None if skip is None else skip.t
        b0 = e["b0"] if trow is None else trow[e["tseg"][0]:e["tseg"][0] + e["tseg"][1]]
        h = _Act(*L.conv2d_nhwc(x.t, e["w0"], m["cout"], 3, 3, bias=b0, src1=s1, coef=coef0, act_in=L.ACT_SILU, want_stats=True))
        coef1 = self._gn([h.stats()], H * W, m["cout"], e["g1"], e["be1"])
        if "wn" in e:
            res = L.conv2d_nhwc(x.t, e["wn"], m["cout"], 1, 1, bias=e["bn"], src1=s1, in_bound=xb)
        else:
            res = x.t
        return _Act(*L.conv2d_nhwc(h.t, e["w1"], m["cout"], 3, 3, bias=e["b1"], coef=coef1, act_in=L.ACT_SILU, res=res,
                                   want_stats=True))

    def _attn(self, key, m, x):
        """AttnBlock.forward (models/unet.py:113-123): one head of width C, logits scaled by 1/sqrt(C)."""
        e = self.w[key]
        B, H, W, C = x.t.shape
        coef = self._gn([x.stats()], H * W, C, e["gamma"], e["beta"])
        qkvb = self._bound_slot(3)
        if qkvb is None:
            ob = None
            qkv = L.conv2d_nhwc(x.t, e["wqkv"], 3 * C, 1, 1, bias=e["bqkv"], coef=coef)
        else:
            qkv, qst = L.conv2d_nhwc(x.t, e["wqkv"], 3 * C, 1, 1, bias=e["bqkv"], coef=coef, want_stats=True)
            L.moments_bound(qst, 0, C, qkvb)
            ob = qkvb[2:3]
        o = L.attention(qkv.view(B, H * W, 3 * C), C, 1, bounds=qkvb)
        return _Act(*L.conv2d_nhwc(o.view(B, H, W, C), e["wo"], C, 1, 1, bias=e["bo"], res=x.t, want_stats=True, in_bound=ob))

    @torch.no_grad()
    def forward_row(self, x, trow, cond=None):
        """x: (B, C*num_frames, H, W) NCHW; trow: this label's bias-table row (None without time conditioning)."""
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
        e0 = self.w[("downblocks", 0)]
        xin = L.pack_nchw_to_nhwc(x, cond, e0["cin_pad"])
        if self.affine_input:
            xin = L.scale_clamp(xin, 2.0, -1.0)          # zero padding channels meet zero weights
        hs = []
        h = None
        for lst, j, m in self.program:
            key = (lst, j)
            if m["kind"] == "conv":
                e = self.w[key]
                src = xin if h is None else h.t
                y = L.conv2d_nhwc(src, e["w"], m["cout"], 3, 3, bias=e["b"])
                if m["stride"] == 2:
                    y = L.upfirdn2d_nhwc(y, ONE, 1, 2, (0, 0))          # keep every 2nd sample of the "same" convolution
                h = _Act(y)
            elif m["kind"] == "res":
                h = self._res(key, m, h, hs.pop() if lst == "upblocks" else None, trow)
            elif m["kind"] == "attn":
                h = self._attn(key, m, h)
            elif m["kind"] == "upsample":
                e = self.w[key]
                up = L.upfirdn2d_nhwc(h.t, BOX2, 2, 1, (1, 0))           # nn.Upsample(scale_factor=2, mode="nearest")
                h = _Act(L.conv2d_nhwc(up, e["w"], m["ch"], 3, 3, bias=e["b"]))
            if lst == "downblocks":
                if m["kind"] == "attn":
                    hs.pop()
                hs.append(h)
        assert not hs
        Bh, Hh, Wh, C = h.t.shape
        coef = self._gn([h.stats()], Hh * Wh, C, self.norm["gamma"], self.norm["beta"])
        co = self.out["co"]
        out = torch.empty((B, H, W, _pad16(co)), device=x.device, dtype=torch.float32)
        L.conv2d_nhwc(h.t, self.out["w"], co, 3, 3, bias=self.out["b"], coef=coef, act_in=L.ACT_SILU, out=out)
        return L.nhwc_to_nchw(out, co)

    def _forward_label(self, x, label, cond):
        self.prepare_labels([label])
        return self.forward_row(x, self._row_bias.get(float(label)), cond)

    def forward_label(self, x, label, cond=None):
        """All samples share one label (what every sampler does)."""
        if self.noise_in_cond and cond is not None:
            cond = self._noised_cond(cond.to(self.device, torch.float32).contiguous(), [float(label)] * x.shape[0])
        return self._forward_label(x, label, cond)

    def __call__(self, x, labels, cond=None, cond_mask=None):
        """Reference call shape ``scorenet(x, labels, cond=cond)``.  The time embedding enters as a per-sample bias of
        conv0; a launch takes one bias vector, so a batch with several distinct labels runs as one launch per label."""
        vals = [float(v) for v in (labels.detach().cpu().tolist() if torch.is_tensor(labels) else labels)]
        assert len(vals) == x.shape[0]
        x = x.to(self.device, torch.float32).contiguous()
        cond = None if cond is None else cond.to(self.device, torch.float32).contiguous()
        cond = self._noised_cond(cond, vals)             # noise_in_cond: per-sample levels, before the batch is regrouped
        uniq = list(dict.fromkeys(vals))
        if len(uniq) == 1 or not self.time_conditional:
            return self._forward_label(x, uniq[0], cond)
        out = torch.empty((x.shape[0], self.out["co"], x.shape[2], x.shape[3]), device=self.device, dtype=torch.float32)
        for v in uniq:
            idx = torch.tensor([i for i, u in enumerate(vals) if u == v], device=self.device)
            out[idx] = self._forward_label(x[idx].contiguous(), v, None if cond is None else cond[idx].contiguous())
        return out

    forward = __call__

    def eval(self):
        return self
