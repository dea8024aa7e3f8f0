"""ELIC key-frame codec on MI355X.

Host-side mirror of the reference's ``TestModel`` (Network.py:74-640) and ``Inference.inference``
(Inference.py:19-75): same state-dict key names, ``compress(x) -> {"strings": [y_strings, z_strings],
"shape"}`` and ``decompress(strings, shape) -> {"x_hat"}`` with the same nested string layout
(``y_strings[slice][anchor|non-anchor][batch]``, ``z_strings[batch]``).

Every convolution runs through ``evc_conv2d_nhwc_f32`` (NHWC, f32 matrix cores):
* ReLUs are folded into the consumer's operand load / the producer's epilogue;
* the 5x5 stride-2 transposed / strided convolutions (compressai ``deconv`` / ``conv``) run in polyphase form
  (``evc_deconv5x5s2_f32`` / ``evc_conv5x5s2_f32``: one 3x3 convolution on the low-resolution side + a layout pass,
  36 instead of 100 MACs per pixel and channel pair; exactly the same sums);
* the checkerboard-masked 5x5 context convolution uses weights masked once at load
  (the reference re-masks on every forward, ELICUtilis/layers/layers.py:85);
* the concat feeding ParamAggregation is never materialised: context + channel-conditional outputs are
  written side by side into one buffer, the hyper-prior output is the second conv source;
* entropy coding: scale -> CDF index, mean gather and symbol scatter are GPU kernels; only int32
  indexes / symbols cross to the host range coder (libevc_rans.so), per (slice, pass) as in the reference.
"""
import numpy as np
import torch

from . import lib as L
from .entropy import EntropyBottleneckCodec, Tables

GROUPS = [0, 16, 16, 32, 64, 192]   # Network.py:87
ONE = np.ones((1, 1), dtype=np.float32)
# Bumped whenever a change to the convolution kernels can alter the last bit of the entropy-parameter networks' outputs
# (summation order, tile shapes ...): streams carry it (container.py) and receivers refuse a mismatch.
ELIC_CODEC_REV = 4
ENTROPY_NETS = ("h_s", "cc_transforms", "context_prediction", "ParamAggregation")


def _pad16(c):
    return (c + 15) // 16 * 16


class ElicModel:
    def __init__(self, state_dict, device="cuda", N=192, M=320):
        L.hip_lib()
        self.device = torch.device(device)
        self.N, self.M = N, M
        self.arith = L.default_arith()          # operand ranges are unknown here: never the fp16 split
        # The entropy-parameter networks (h_s, cc_transforms, context_prediction, ParamAggregation: everything between the decoded
        # integers and the means / scales the range coder is driven with) ALWAYS run under EVC_ARITH_F32: every output is one
        # fixed-order chain of fmaf (v_mfma_f32_32x32x2_f32), which a CPU reproduces bit for bit (oracle/exact_conv.c).  A stream
        # therefore decodes to the same symbols on any implementation of that arithmetic -- not only on this kernel build --
        # which is what "bit-exact integer symbols" needs across implementations (tests/test_gpu_elic.py).  g_a / h_a / g_s are
        # not part of that contract and keep the faster split arithmetic.
        self.entropy_arith = L.ARITH_F32
        sd = state_dict
        self.w = {}
        for k in sd:
            if not k.endswith(".weight"):
                continue
            name = k[:-len(".weight")]
            if not name.split(".")[0] in ("g_a", "g_s", "h_a", "h_s", "cc_transforms", "context_prediction",
                                          "ParamAggregation"):
                continue
            w = sd[k].detach().float()
            arith = self.entropy_arith if name.split(".")[0] in ENTROPY_NETS else self.arith
            if self._is_deconv(name):                         # compressai deconv(): polyphase 3x3 form
                self.w[name] = L.Deconv5x5s2(w, sd[name + ".bias"], arith, self.device)
                continue
            if self._is_conv_s2(name):                        # compressai conv(): space-to-depth + 3x3
                self.w[name] = L.Conv5x5s2(w, sd[name + ".bias"], self.arith, self.device)
                continue
            if name.startswith("context_prediction"):
                mask = sd.get(name + ".mask")
                if mask is None:
                    mask = torch.zeros_like(w)
                    mask[:, :, 0::2, 1::2] = 1
                    mask[:, :, 1::2, 0::2] = 1
                w = w * mask.float()
            co, ci, kh, kw = w.shape
            cip = _pad16(ci)
            if cip != ci:
                wp = torch.zeros(co, cip, kh, kw)
                wp[:, :ci] = w
                w = wp
            self.w[name] = dict(w=L.conv_pack_weights(w.to(self.device), arith), b=sd[name + ".bias"].detach().float().to(self.device),
                                co=co, k=kh)
        self.gc = Tables.from_state_dict(sd, "gaussian_conditional")
        self.scale_table = sd["gaussian_conditional.scale_table"].detach().float().to(self.device).contiguous()
        self.eb = EntropyBottleneckCodec(Tables.from_state_dict(sd, "entropy_bottleneck"),
                                         sd["entropy_bottleneck.quantiles"][:, 0, 1].detach().float().cpu().numpy())

    def codec_tag(self):
        """(convolution arithmetic, revision) the entropy parameters of this model are computed with."""
        return (self.entropy_arith, ELIC_CODEC_REV)

    @staticmethod
    def _is_deconv(name):
        return name in ("g_s.1", "g_s.5", "g_s.10", "g_s.14", "h_s.0", "h_s.2")

    @staticmethod
    def _is_conv_s2(name):
        return name in ("g_a.0", "g_a.4", "g_a.9", "g_a.13", "h_a.2", "h_a.4")

    # ---- layer helpers (all NHWC) --------------------------------------------------------------
    def _conv(self, name, x, src1=None, act_in=L.ACT_NONE, act_out=L.ACT_NONE, res=None, out=None):
        e = self.w[name]
        # splits=1: no split-K, so every output element is one fixed-order sum whatever the batch size --
        # encoder and decoder (possibly run with different batch compositions) see bit-identical
        # means / scales, which entropy decoding needs.
        return L.conv2d_nhwc(x, e["w"], e["co"], e["k"], e["k"], bias=e["b"], src1=src1, act_in=act_in,
                             act_out=act_out, res=res, out=out, splits=1)

    def _deconv(self, name, x, act_out=L.ACT_NONE):
        return self.w[name](x, act_out=act_out)                   # (B, H, W, Ci) -> (B, 2H, 2W, Co)

    def _conv_s2(self, name, x, act_out=L.ACT_NONE):
        op = self.w[name]
        return op(x, act_out=act_out, channels=op.Ci)             # (B, 2H, 2W, ld >= Ci) -> (B, H, W, Co)

    def _rbb(self, n, x):
        """ResidualBottleneckBlock (Network.py:48-59)."""
        t = self._conv(n + ".conv1", x, act_out=L.ACT_RELU)
        t = self._conv(n + ".conv2", t, act_out=L.ACT_RELU)
        return self._conv(n + ".conv3", t, res=x)

    def _res_unit(self, n, x):
        t = self._conv(n + ".conv.0", x, act_out=L.ACT_RELU)
        t = self._conv(n + ".conv.2", t, act_out=L.ACT_RELU)
        return self._conv(n + ".conv.4", t, res=x, act_out=L.ACT_RELU)

    def _attention(self, n, x):
        """AttentionBlock (ELICUtilis/layers/layers.py:202-253)."""
        a = x
        for i in range(3):
            a = self._res_unit(f"{n}.conv_a.{i}", a)
        b = x
        for i in range(3):
            b = self._res_unit(f"{n}.conv_b.{i}", b)
        b = self._conv(n + ".conv_b.3", b)
        return L.gate_residual(a, b, x)

    def g_s(self, y):
        x = self._attention("g_s.0", y)
        x = self._deconv("g_s.1", x)
        for i in (2, 3, 4):
            x = self._rbb(f"g_s.{i}", x)
        x = self._deconv("g_s.5", x)
        x = self._attention("g_s.6", x)
        for i in (7, 8, 9):
            x = self._rbb(f"g_s.{i}", x)
        x = self._deconv("g_s.10", x)
        for i in (11, 12, 13):
            x = self._rbb(f"g_s.{i}", x)
        return self._deconv("g_s.14", x)          # (B, H, W, 3)

    def g_a(self, x):
        x = self._conv_s2("g_a.0", x)
        for i in (1, 2, 3):
            x = self._rbb(f"g_a.{i}", x)
        x = self._conv_s2("g_a.4", x)
        for i in (5, 6, 7):
            x = self._rbb(f"g_a.{i}", x)
        x = self._attention("g_a.8", x)
        x = self._conv_s2("g_a.9", x)
        for i in (10, 11, 12):
            x = self._rbb(f"g_a.{i}", x)
        x = self._conv_s2("g_a.13", x)
        return self._attention("g_a.14", x)

    def h_a(self, y):
        x = self._conv("h_a.0", y, act_out=L.ACT_RELU)
        x = self._conv_s2("h_a.2", x, act_out=L.ACT_RELU)
        return self._conv_s2("h_a.4", x)

    def h_s(self, z):
        x = self._deconv("h_s.0", z, act_out=L.ACT_RELU)
        x = self._deconv("h_s.2", x, act_out=L.ACT_RELU)
        return self._conv("h_s.4", x)               # (B, H, W, 2M) = [means | scales]

    # ---- the slice / checkerboard loop shared by encode and decode -------------------------------
    def _slice_loop(self, hs, y_hat, code_pass):
        """hs: (B,H,W,2M) hyper-prior output; y_hat: (B,H,W,M) zero-initialised, filled in place.
        ``code_pass(i, parity, c0, g, idx, means)`` returns the int32 symbols (B,g,H,W/2) on the device."""
        B, H, W, _ = hs.shape
        c0 = 0
        prev = None
        for i in range(len(GROUPS) - 1):
            g = GROUPS[i + 1]
            wide = 2 * g if i == 0 else 4 * g
            pa_in = torch.zeros((B, H, W, wide), device=self.device, dtype=torch.float32)
            if i > 0:
                s0 = L.Cols(y_hat, 0, GROUPS[1])
                s1 = None if i == 1 else L.Cols(y_hat, prev[0], prev[1])
                t = self._conv(f"cc_transforms.{i - 1}.0", s0, src1=s1, act_out=L.ACT_RELU)
                t = self._conv(f"cc_transforms.{i - 1}.2", t, act_out=L.ACT_RELU)
                self._conv(f"cc_transforms.{i - 1}.4", t, out=L.Cols(pa_in, 2 * g, 2 * g))
            for parity in (0, 1):
                if parity == 1:   # masked context of the decoded anchors (non-anchor sites are still zero)
                    self._conv(f"context_prediction.{i}", L.Cols(y_hat, c0, g), out=L.Cols(pa_in, 0, 2 * g))
                t = self._conv(f"ParamAggregation.{i}.0", pa_in, src1=hs, act_out=L.ACT_RELU)
                t = self._conv(f"ParamAggregation.{i}.2", t, act_out=L.ACT_RELU)
                ms = self._conv(f"ParamAggregation.{i}.4", t)                    # (B,H,W,2g) = [means | scales]
                idx, means = L.elic_gather_params(ms, 0, g, g, parity, self.scale_table)
                sym = code_pass(i, parity, c0, g, idx, means)
                L.elic_scatter_symbols(sym, means, y_hat, c0, parity)
            prev = (c0, g)
            c0 += g

    # ---- public API --------------------------------------------------------------------------------
    @torch.no_grad()
    def decompress(self, strings, shape, return_latents=False):
        """TestModel.decompress (Network.py:444-532) for a batch: strings = [y_strings, z_strings]."""
        assert isinstance(strings, list) and len(strings) == 2
        y_strings, z_strings = strings
        B = len(z_strings)
        z_hat = torch.from_numpy(self.eb.decompress(z_strings, tuple(shape))).to(self.device)
        hs = self.h_s(z_hat.permute(0, 2, 3, 1).contiguous())
        H, W = shape[0] * 4, shape[1] * 4
        y_hat = torch.zeros((B, H, W, self.M), device=self.device, dtype=torch.float32)
        all_syms = []

        def decode_pass(i, parity, c0, g, idx, means):
            idx_h = idx.cpu().numpy()            # device -> host: int32 CDF indexes
            sym = np.empty(idx_h.shape, dtype=np.int32)
            for b in range(B):
                sym[b] = self.gc.decode(y_strings[i][parity][b], idx_h[b].reshape(-1)).reshape(idx_h[b].shape)
            if return_latents:
                all_syms.append(sym.copy())
            return torch.from_numpy(sym).to(self.device)
        self._slice_loop(hs, y_hat, decode_pass)
        x = self.g_s(y_hat)
        x_hat = L.scale_clamp(L.nhwc_to_nchw(x, 3), 1.0, 0.0, (0.0, 1.0))      # .clamp_(0, 1)
        out = {"x_hat": x_hat}
        if return_latents:
            out.update(y_hat=y_hat.permute(0, 3, 1, 2).contiguous(), z_hat=z_hat, symbols=all_syms)
        return out

    @torch.no_grad()
    def compress(self, x, return_latents=False):
        """TestModel.compress (Network.py:336-441); x: (B, 3, H, W) in [0, 1], H and W multiples of 64."""
        x = x.to(self.device, torch.float32).contiguous()
        B = x.shape[0]
        y = self.g_a(L.pack_nchw_to_nhwc(x, None, 16))
        z = self.h_a(y)
        z_nchw = z.permute(0, 3, 1, 2).contiguous().cpu().numpy()
        z_strings = self.eb.compress(z_nchw)
        shape = z_nchw.shape[-2:]
        z_hat = torch.from_numpy(self.eb.decompress(z_strings, shape)).to(self.device)
        hs = self.h_s(z_hat.permute(0, 2, 3, 1).contiguous())
        y_hat = torch.zeros_like(y)
        y_strings = [[None, None] for _ in range(len(GROUPS) - 1)]

        def encode_pass(i, parity, c0, g, idx, means):
            sym = L.elic_quantize(y, c0, means, parity)
            sym_h, idx_h = sym.cpu().numpy(), idx.cpu().numpy()
            y_strings[i][parity] = [self.gc.encode(sym_h[b].reshape(-1), idx_h[b].reshape(-1)) for b in range(B)]
            return sym
        self._slice_loop(hs, y_hat, encode_pass)
        out = {"strings": [y_strings, z_strings], "shape": tuple(int(s) for s in shape)}
        if return_latents:
            out.update(y=y.permute(0, 3, 1, 2).contiguous(), y_hat=y_hat.permute(0, 3, 1, 2).contiguous())
        return out


def count_bits(strings):
    """Inference.py:51-67: 8 x total byte length of the nested [y_strings, z_strings] lists."""
    total = 0
    for s in strings:
        for j in s:
            if isinstance(j, list):
                for i in j:
                    total += sum(len(k) for k in i) if isinstance(i, list) else len(i)
            else:
                total += len(j)
    return 8 * total


@torch.no_grad()
def inference(model, x, patch=64):
    """Inference.inference (Inference.py:19-75): x (3,H,W) in [0,1] -> (x_hat (1,3,H,W), bits)."""
    x = x.unsqueeze(0)
    h, w = x.size(2), x.size(3)
    new_h, new_w = (h + patch - 1) // patch * patch, (w + patch - 1) // patch * patch
    xp = torch.nn.functional.pad(x, (0, new_w - w, 0, new_h - h))
    enc = model.compress(xp)
    dec = model.decompress(enc["strings"], enc["shape"])
    x_hat = dec["x_hat"][:, :, :h, :w]
    return x_hat, count_bits(enc["strings"])
