"""On-disk container for one batch of compressed clips.

The reference never serialises anything: key-frame ``strings`` live in RAM and only their bit count is used
(Inference.py:51-67, city_sender.py:556-558).  A receiver needs exactly two things per clip: the transmit mask
``d`` (which frames are key frames) and, per key frame, the ELIC strings ``[y_strings[5][2], z_string]`` plus the
hyper-latent shape.  Layout (little endian):

    magic "EVC1" | u8 format (2) | u8 arith | u16 codec_rev
                 | u16 frames | u16 n_clips | u16 n_key | u16 shape_h | u16 shape_w | u8 d[frames]
    then for every key frame k, clip b:  u32 len(z) | z | for slice 0..4, pass 0..1:  u32 len | bytes

``arith`` / ``codec_rev`` name the arithmetic the ENCODER's entropy-parameter networks ran with
(``ElicModel.codec_tag()``: the convolution arithmetic EVC_ARITH_* and a revision number bumped whenever a kernel's
summation order changes).  A learned codec's range decoder desynchronises if a predicted scale differs by one ulp, so
a receiver whose networks run another arithmetic must REFUSE the stream (``CodecMismatch``) rather than decode garbage.

The payload bytes are exactly the strings the codec produced, so ``8 * payload`` equals the reference's bit count.
"""
import struct

import numpy as np

MAGIC = b"EVC1"
FORMAT = 2
HEADER_BYTES = 4 + 4 + 10          # magic + (format, arith, codec_rev) + 5 x u16
N_SLICES, N_PASSES = 5, 2
ARITH_NAMES = {0: "f32", 1: "bf16x6", 2: "f16x3"}


class CodecMismatch(ValueError):
    """The stream was produced by entropy-parameter networks running a different arithmetic / kernel revision."""


def pack(d, key_strings, shape, codec=(1, 1)):
    """d: (frames,) 0/1; key_strings: list over key frames of [y_strings[5][2][B], z_strings[B]];
    codec: ``ElicModel.codec_tag()`` of the encoder = (arith, codec_rev)."""
    d = np.asarray(d, dtype=np.uint8).reshape(-1)
    n_key = len(key_strings)
    n_clips = len(key_strings[0][1]) if n_key else 0
    assert int(d.sum()) == n_key, "mask and key-frame count disagree"
    out = [MAGIC, struct.pack("<BBH", FORMAT, int(codec[0]), int(codec[1])),
           struct.pack("<5H", len(d), n_clips, n_key, int(shape[0]), int(shape[1])), d.tobytes()]
    for ys, zs in key_strings:
        for b in range(n_clips):
            out.append(struct.pack("<I", len(zs[b])) + zs[b])
            for i in range(N_SLICES):
                for p in range(N_PASSES):
                    s = ys[i][p][b]
                    out.append(struct.pack("<I", len(s)) + s)
    return b"".join(out)


def read_codec(blob):
    """-> (arith, codec_rev) recorded by the encoder."""
    if blob[:4] != MAGIC or len(blob) < HEADER_BYTES:
        raise ValueError("not an EVC1 container")
    fmt, arith, rev = struct.unpack_from("<BBH", blob, 4)
    if fmt != FORMAT:
        raise ValueError(f"unsupported EVC1 container format {fmt} (this build reads format {FORMAT})")
    return arith, rev


def unpack(blob, expect_codec=None):
    """-> (d, key_strings, shape) in the structure ``ClipDecoder.decode`` takes.  ``expect_codec`` = the receiver's
    ``ElicModel.codec_tag()``: a stream coded under another arithmetic / kernel revision raises ``CodecMismatch``."""
    codec = read_codec(blob)
    if expect_codec is not None and tuple(codec) != tuple(int(v) for v in expect_codec):
        raise CodecMismatch(f"stream was coded with convolution arithmetic {ARITH_NAMES.get(codec[0], codec[0])} rev {codec[1]}, "
                            f"this receiver runs {ARITH_NAMES.get(int(expect_codec[0]), expect_codec[0])} rev {int(expect_codec[1])}: "
                            "the entropy parameters would differ in the last bit and the range decoder would "
                            "desynchronise (set EVC_CONV_ARITH to match the sender)")
    frames, n_clips, n_key, sh, sw = struct.unpack_from("<5H", blob, 8)
    off = HEADER_BYTES
    d = np.frombuffer(blob, dtype=np.uint8, count=frames, offset=off).astype(np.int64)
    off += frames
    if int(d.sum()) != n_key:
        raise ValueError("corrupt container: mask and key-frame count disagree")

    def take():
        nonlocal off
        if off + 4 > len(blob):
            raise ValueError("truncated container")
        (n,) = struct.unpack_from("<I", blob, off)
        off += 4
        if off + n > len(blob):
            raise ValueError("truncated container")
        s = bytes(blob[off:off + n])
        off += n
        return s
    key_strings = []
    for _ in range(n_key):
        ys = [[[None] * n_clips for _ in range(N_PASSES)] for _ in range(N_SLICES)]
        zs = [None] * n_clips
        for b in range(n_clips):
            zs[b] = take()
            for i in range(N_SLICES):
                for p in range(N_PASSES):
                    ys[i][p][b] = take()
        key_strings.append([ys, zs])
    if off != len(blob):
        raise ValueError("trailing bytes in container")
    return d, key_strings, (sh, sw)


def payload_bits(key_strings):
    return 8 * sum(len(z) for ys, zs in key_strings for z in zs) + \
        8 * sum(len(s) for ys, zs in key_strings for sl in ys for ps in sl for s in ps)
