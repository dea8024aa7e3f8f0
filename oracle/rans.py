"""Oracle: range-ANS coder in plain Python integers (test infrastructure only; small cases).

PARITY UNPINNED: the algorithm lives in third-party compressai==1.1.5 (reference requirements.txt:17;
not vendored, not installed).  This restates its published `rans_interface.cpp` / ryg_rans `rans64.h`:
64-bit state, lower bound L = 2^31, 32-bit renormalisation words, 16-bit CDF precision, 4-bit bypass
chunks for out-of-range values.  Reference call sites: Network.py:346-347, 400-401, 424-428 (encode),
Network.py:450, 495, 516 (decode).  It shares no code with csrc/rans.cpp.
"""
import struct

PRECISION = 16
BYPASS = 4
MAX_BYPASS = (1 << BYPASS) - 1
RANS_L = 1 << 31


def _items(symbols, indexes, cdfs, cdf_sizes, offsets):
    """Forward list of (start, range, is_bypass) items, in the order the decoder will see them."""
    items = []
    for sym, ci in zip(symbols, indexes):
        cdf = cdfs[ci]
        max_value = int(cdf_sizes[ci]) - 2
        value = int(sym) - int(offsets[ci])
        raw = 0
        if value < 0:
            raw = -2 * value - 1
            value = max_value
        elif value >= max_value:
            raw = 2 * (value - max_value)
            value = max_value
        items.append((int(cdf[value]), int(cdf[value + 1]) - int(cdf[value]), False))
        if value == max_value:
            n_bypass = 0
            while (raw >> (n_bypass * BYPASS)) != 0:
                n_bypass += 1
            val = n_bypass
            while val >= MAX_BYPASS:
                items.append((MAX_BYPASS, MAX_BYPASS + 1, True))
                val -= MAX_BYPASS
            items.append((val, val + 1, True))
            for j in range(n_bypass):
                v = (raw >> (j * BYPASS)) & MAX_BYPASS
                items.append((v, v + 1, True))
    return items


def encode_with_indexes(symbols, indexes, cdfs, cdf_sizes, offsets):
    words = []          # emitted in reverse stream order
    x = RANS_L
    for start, rng, bypass in reversed(_items(symbols, indexes, cdfs, cdf_sizes, offsets)):
        if not bypass:
            x_max = ((RANS_L >> PRECISION) << 32) * rng
            if x >= x_max:
                words.append(x & 0xFFFFFFFF)
                x >>= 32
            x = ((x // rng) << PRECISION) + (x % rng) + start
        else:
            freq = 1 << (16 - BYPASS)
            x_max = ((RANS_L >> 16) << 32) * freq
            if x >= x_max:
                words.append(x & 0xFFFFFFFF)
                x >>= 32
            x = (x << BYPASS) | start
    words.append((x >> 32) & 0xFFFFFFFF)
    words.append(x & 0xFFFFFFFF)
    words.reverse()
    return struct.pack("<%dI" % len(words), *words)


def decode_with_indexes(data, indexes, cdfs, cdf_sizes, offsets):
    words = struct.unpack("<%dI" % (len(data) // 4), data)
    pos = 2
    x = words[0] | (words[1] << 32)
    mask = (1 << PRECISION) - 1

    def get_bits():
        nonlocal x, pos
        val = x & ((1 << BYPASS) - 1)
        x >>= BYPASS
        if x < RANS_L:
            x = (x << 32) | words[pos]
            pos += 1
        return val

    out = []
    for ci in indexes:
        cdf = cdfs[ci]
        size = int(cdf_sizes[ci])
        max_value = size - 2
        cum = x & mask
        s = 0
        while s < size and cdf[s] <= cum:
            s += 1
        s -= 1
        start, freq = int(cdf[s]), int(cdf[s + 1]) - int(cdf[s])
        x = freq * (x >> PRECISION) + (x & mask) - start
        if x < RANS_L:
            x = (x << 32) | words[pos]
            pos += 1
        value = s
        if value == max_value:
            val = get_bits()
            n_bypass = val
            while val == MAX_BYPASS:
                val = get_bits()
                n_bypass += val
            raw = 0
            for j in range(n_bypass):
                raw |= get_bits() << (j * BYPASS)
            value = raw >> 1
            value = -value - 1 if raw & 1 else value + max_value
        out.append(value + int(offsets[ci]))
    return out
