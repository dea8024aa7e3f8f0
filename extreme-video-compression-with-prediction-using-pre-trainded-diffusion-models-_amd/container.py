"""On-disk container for one batch of compressed clips.

The reference never serialises anything: key-frame ``strings`` live in RAM and only their bit count is used
(Inference.py:51-67, city_sender.py:556-558).  A receiver needs exactly two things per clip: the transmit mask
``d`` (which frames are key frames) and, per key frame, the ELIC strings ``[y_strings[5][2], z_string]`` plus the
hyper-latent shape.  Layout (little endian):

    magic "EVC1" | u16 frames | u16 n_clips | u16 n_key | u16 shape_h | u16 shape_w | u8 d[frames]
    then for every key frame k, clip b:  u32 len(z) | z | for slice 0..4, pass 0..1:  u32 len | bytes

The payload bytes are exactly the strings the codec produced, so ``8 * payload`` equals the reference's bit count.
"""
import struct

import numpy as np

MAGIC = b"EVC1"
N_SLICES, N_PASSES = 5, 2


def pack(d, key_strings, shape):
    """d: (frames,) 0/1; key_strings: list over key frames of [y_strings[5][2][B], z_strings[B]]."""
    d = np.asarray(d, dtype=np.uint8).reshape(-1)
    n_key = len(key_strings)
    n_clips = len(key_strings[0][1]) if n_key else 0
    assert int(d.sum()) == n_key, "mask and key-frame count disagree"
    out = [MAGIC, struct.pack("<5H", len(d), n_clips, n_key, int(shape[0]), int(shape[1])), d.tobytes()]
    for ys, zs in key_strings:
        for b in range(n_clips):
            out.append(struct.pack("<I", len(zs[b])) + zs[b])
            for i in range(N_SLICES):
                for p in range(N_PASSES):
                    s = ys[i][p][b]
                    out.append(struct.pack("<I", len(s)) + s)
    return b"".join(out)


def unpack(blob):
    """-> (d, key_strings, shape) in the structure ``ClipDecoder.decode`` takes."""
    if blob[:4] != MAGIC:
        raise ValueError("not an EVC1 container")
    frames, n_clips, n_key, sh, sw = struct.unpack_from("<5H", blob, 4)
    off = 14
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
