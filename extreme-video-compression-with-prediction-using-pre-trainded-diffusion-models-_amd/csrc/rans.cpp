// rans.cpp -- host range-ANS coder behind include/evc_rans.h (plain C++17, no GPU, no torch).
#include "../../include/evc_rans.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <vector>

namespace {

constexpr int kPrecision = 16;
constexpr int kBypassBits = 4;
constexpr int kMaxBypass = (1 << kBypassBits) - 1;
constexpr uint64_t kRansL = 1ull << 31;   // lower bound of the normalised state interval

struct Sym { uint16_t start; uint16_t range; bool bypass; };

inline void enc_put(uint64_t& x, uint32_t*& ptr, uint32_t start, uint32_t freq, uint32_t scale_bits) {
    const uint64_t x_max = ((kRansL >> scale_bits) << 32) * freq;
    if (x >= x_max) { *--ptr = (uint32_t)x; x >>= 32; }
    x = ((x / freq) << scale_bits) + (x % freq) + start;
}

inline void enc_put_bits(uint64_t& x, uint32_t*& ptr, uint32_t val, uint32_t nbits) {
    const uint32_t freq = 1u << (16 - nbits);
    const uint64_t x_max = ((kRansL >> 16) << 32) * freq;
    if (x >= x_max) { *--ptr = (uint32_t)x; x >>= 32; }
    x = (x << nbits) | val;
}

struct Reader {
    const uint8_t* p; const uint8_t* end; bool bad = false;
    uint32_t next() {
        if (p + 4 > end) { bad = true; return 0; }
        uint32_t w; std::memcpy(&w, p, 4); p += 4; return w;
    }
};

}  // namespace

extern "C" const char* evc_rans_version(void) { return "evc-rans 0.1 (rans64, precision 16, bypass 4)"; }

extern "C" long long evc_rans_max_encoded_bytes(long long n) {
    // one 32-bit word per coded item is the hard ceiling (each put emits at most one word); a symbol in
    // bypass mode adds 1 + ceil(32/4) + small prefix items; 8 bytes of final state.
    return n < 0 ? EVC_RANS_EINVAL : (n * 24 + 4) * 4 + 8;
}

extern "C" long long evc_rans_encode_with_indexes(const int32_t* symbols, const int32_t* indexes, long long n,
                                                  const int32_t* cdfs, int cdf_ld, const int32_t* cdf_sizes,
                                                  const int32_t* offsets, int n_cdfs, uint8_t* out,
                                                  long long out_cap) {
    if (n < 0 || (n > 0 && (!symbols || !indexes)) || !cdfs || !cdf_sizes || !offsets || !out || n_cdfs <= 0)
        return EVC_RANS_EINVAL;
    std::vector<Sym> syms;
    syms.reserve((size_t)n + 16);
    for (long long i = 0; i < n; ++i) {
        const int32_t ci = indexes[i];
        if (ci < 0 || ci >= n_cdfs) return EVC_RANS_EINVAL;
        const int32_t* cdf = cdfs + (size_t)ci * cdf_ld;
        const int32_t max_value = cdf_sizes[ci] - 2;
        if (max_value < 0 || cdf_sizes[ci] > cdf_ld) return EVC_RANS_EINVAL;
        int32_t value = symbols[i] - offsets[ci];
        uint32_t raw = 0;
        if (value < 0) { raw = (uint32_t)(-2 * (int64_t)value - 1); value = max_value; }
        else if (value >= max_value) { raw = (uint32_t)(2 * ((int64_t)value - max_value)); value = max_value; }
        const int32_t lo = cdf[value], hi = cdf[value + 1];
        if (hi <= lo) return EVC_RANS_EINVAL;   // zero-frequency symbol cannot be coded
        syms.push_back({(uint16_t)lo, (uint16_t)(hi - lo), false});
        if (value == max_value) {
            int32_t n_bypass = 0;
            while ((raw >> (n_bypass * kBypassBits)) != 0) ++n_bypass;
            int32_t val = n_bypass;
            while (val >= kMaxBypass) { syms.push_back({(uint16_t)kMaxBypass, (uint16_t)(kMaxBypass + 1), true}); val -= kMaxBypass; }
            syms.push_back({(uint16_t)val, (uint16_t)(val + 1), true});
            for (int32_t j = 0; j < n_bypass; ++j) {
                const uint32_t v = (raw >> (j * kBypassBits)) & kMaxBypass;
                syms.push_back({(uint16_t)v, (uint16_t)(v + 1), true});
            }
        }
    }
    std::vector<uint32_t> buf(syms.size() + 2);
    uint32_t* ptr = buf.data() + buf.size();
    uint64_t x = kRansL;
    for (size_t i = syms.size(); i-- > 0;) {
        const Sym& s = syms[i];
        if (!s.bypass) enc_put(x, ptr, s.start, s.range, kPrecision);
        else enc_put_bits(x, ptr, s.start, kBypassBits);
    }
    ptr -= 2;
    ptr[0] = (uint32_t)x; ptr[1] = (uint32_t)(x >> 32);
    const long long nbytes = (long long)(buf.data() + buf.size() - ptr) * 4;
    if (nbytes > out_cap) return EVC_RANS_ENOSPC;
    std::memcpy(out, ptr, (size_t)nbytes);
    return nbytes;
}

extern "C" int evc_rans_decode_with_indexes(const uint8_t* in, long long n_bytes, const int32_t* indexes, long long n,
                                            const int32_t* cdfs, int cdf_ld, const int32_t* cdf_sizes,
                                            const int32_t* offsets, int n_cdfs, int32_t* symbols) {
    if (!in || n_bytes < 8 || n < 0 || (n > 0 && (!indexes || !symbols)) || !cdfs || !cdf_sizes || !offsets ||
        n_cdfs <= 0)
        return EVC_RANS_EINVAL;
    Reader rd{in, in + n_bytes};
    uint64_t x = rd.next();
    x |= (uint64_t)rd.next() << 32;
    const uint64_t mask = (1ull << kPrecision) - 1;
    auto get_bits = [&](uint32_t nbits) {
        const uint32_t val = (uint32_t)(x & ((1u << nbits) - 1));
        x >>= nbits;
        if (x < kRansL) x = (x << 32) | rd.next();
        return (int32_t)val;
    };
    for (long long i = 0; i < n; ++i) {
        const int32_t ci = indexes[i];
        if (ci < 0 || ci >= n_cdfs) return EVC_RANS_EINVAL;
        const int32_t* cdf = cdfs + (size_t)ci * cdf_ld;
        const int32_t size = cdf_sizes[ci];
        if (size < 2 || size > cdf_ld) return EVC_RANS_EINVAL;
        const int32_t max_value = size - 2;
        const uint32_t cum = (uint32_t)(x & mask);
        // first entry > cum, minus one (tables are short: linear scan as upstream does)
        int32_t s = 0;
        while (s < size && (uint32_t)cdf[s] <= cum) ++s;
        s -= 1;
        if (s < 0 || s > max_value) return EVC_RANS_ECORRUPT;
        const uint32_t start = (uint32_t)cdf[s], freq = (uint32_t)(cdf[s + 1] - cdf[s]);
        x = freq * (x >> kPrecision) + (x & mask) - start;
        if (x < kRansL) x = (x << 32) | rd.next();
        int32_t value = s;
        if (value == max_value) {
            int32_t val = get_bits(kBypassBits);
            int32_t n_bypass = val;
            while (val == kMaxBypass) { val = get_bits(kBypassBits); n_bypass += val; if (rd.bad) return EVC_RANS_ECORRUPT; }
            if (n_bypass > 8) return EVC_RANS_ECORRUPT;   // raw value is at most 32 bits
            uint32_t raw = 0;
            for (int32_t j = 0; j < n_bypass; ++j) raw |= (uint32_t)get_bits(kBypassBits) << (j * kBypassBits);
            value = (int32_t)(raw >> 1);
            if (raw & 1) value = -value - 1;
            else value += max_value;
        }
        if (rd.bad) return EVC_RANS_ECORRUPT;
        symbols[i] = value + offsets[ci];
    }
    return 0;
}

extern "C" int evc_pmf_to_quantized_cdf(const float* pmf, int n, int precision, int32_t* out) {
    if (!pmf || !out || n <= 0 || precision <= 0 || precision > 16) return EVC_RANS_EINVAL;
    std::vector<uint32_t> cdf((size_t)n + 1);
    cdf[0] = 0;
    for (int i = 0; i < n; ++i) cdf[i + 1] = (uint32_t)std::lround(pmf[i] * (float)(1 << precision));
    const uint32_t total = std::accumulate(cdf.begin(), cdf.end(), 0u);
    if (total == 0) return EVC_RANS_EINVAL;
    for (auto& p : cdf) p = (uint32_t)((((uint64_t)1 << precision) * p) / total);
    std::partial_sum(cdf.begin(), cdf.end(), cdf.begin());
    cdf.back() = 1u << precision;
    for (int i = 0; i < n; ++i) {
        if (cdf[i] == cdf[i + 1]) {
            uint32_t best_freq = ~0u; int best = -1;   // steal from the lowest-frequency symbol with freq > 1
            for (int j = 0; j < n; ++j) {
                const uint32_t f = cdf[j + 1] - cdf[j];
                if (f > 1 && f < best_freq) { best_freq = f; best = j; }
            }
            if (best < 0) return EVC_RANS_EINVAL;
            if (best < i) for (int j = best + 1; j <= i; ++j) cdf[j]--;
            else for (int j = i + 1; j <= best; ++j) cdf[j]++;
        }
    }
    for (int i = 0; i <= n; ++i) out[i] = (int32_t)cdf[i];
    return 0;
}
