// conv_igemm.hip -- stride-1 "same" convolution as an implicit GEMM on the gfx950 f32 matrix cores.
//
//   D[m][co] = sum_{tap, ci} act(src[m shifted by tap][ci] * a[b][ci] + s[b][ci]) * W[co][ci][tap]
//
// GEMM view: M = B*H*W output pixels (A operand, gathered from NHWC activations), N = Co (B operand,
// pre-packed weights), K = KH*KW*Ci walked in steps of KC = 16 channels of one tap.
// One workgroup = 4 waves (2 x 2) computes a 128 x (64*TN) tile with v_mfma_f32_32x32x2_f32 (exact f32:
// bitwise a k-ordered fmaf chain); each wave owns 64 pixels x 32*TN channels (6 accumulator tiles at TN=3).
//
// Per K-step, for the NEXT step:
//   * W slab [64*TN][16] goes global -> LDS by LDS-DMA (global_load_lds_dwordx4): no VGPRs, no ds_write;
//     the packed weights are stored pre-swizzled, so a linear copy lands in the conflict-free layout;
//   * A tile [128][16] goes global -> registers (2 float4 per thread), is transformed in registers
//     (GroupNorm affine + SiLU/ReLU; out-of-image taps are zeroed by a select AFTER the activation,
//     which is what zero padding of the activated tensor means) and written to LDS between the second
//     half of the MFMAs.  Addresses are 32-bit offsets advanced incrementally (no divisions in the loop);
//     tap validity is a per-thread bit mask built once; affine coefficients reload only when the channel
//     chunk changes.
// The K-step is straight-line code with one barrier.  LDS tiles are unpadded 64-byte rows with the 16-byte
// chunk index XOR-ed by (row >> 2) & 3: every ds_read_b128 fragment read (lanes 0-31 take k 0..3, lanes 32-63
// k 4..7 of each 8-deep group) and every ds_write_b128 is bank-conflict free.  40 KB LDS per workgroup.
//
// Measured on MI355X (tools/conv_bench.hip): the MFMA + fragment-read + barrier skeleton alone reaches ~130
// TFLOP/s at the ~2.18 GHz the chip holds under this load (fp32 roof 157.3 at 2.4 GHz); every producer
// instruction costs issue slots, which is why the producer side is kept this thin.
//
// Replaces nn.Conv2d 3x3/1x1 (reference models/better/layers.py:89-113), NIN (layers.py:535-544),
// nn.Linear (ncsnpp_more.py:89-95, layerspp.py:507) and the ELIC conv stacks (Network.py:106-166).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/evc_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

#ifndef EVC_CONV_TAIL
#define EVC_CONV_TAIL 1        // row-reuse kernel, unsplit grids of 1.x / 2.x rounds: the tiles of the last partial round are split along K
                               // so that they fill the machine once more with short jobs (0 = off, for A/B; run-time option "tail_split")
#endif
#ifndef EVC_CONV_TM1
#define EVC_CONV_TM1 1         // f32 kernel: grids with < 64 tiles of 128 pixels use 64-pixel tiles (0 disables, for A/B)
#endif
#ifndef EVC_SPLIT_WIDE_TILES
#define EVC_SPLIT_WIDE_TILES 1 // 256-pixel / 8-wave form of the row-reuse kernel on grids of whole rounds (run-time option "wide_tiles")
#endif

#ifndef EVC_SPLIT_INTERLEAVE
#define EVC_SPLIT_INTERLEAVE 5 // bf16x6 pipelined kernel: VALU instructions scheduled per MFMA in the second half (0 = compiler's order)
#endif
// Variants that were measured and did not pay -- 2-D patch tiles, a software-pipelined row-reuse loop, staging balance /
// interleave switches, XCD-contiguous tile order, LDS-DMA of plain activation tiles, producer / consumer specialised kernels,
// an in-kernel ("last-arriver") split-K combine, weight DMA issued mid-step, a ping-pong schedule of the 8-wave tile, a static
// priority for one of the two workgroups of a CU, the ablation diagnostics behind profiles/r02_conv_bench_ablation*.log --
// live in tools/experiments/, outside this file.

namespace {

constexpr int BM_MAX = 128;  // pixels per workgroup tile: 64 * TM (TM = 32-row MFMA tiles per wave, 1 or 2)
constexpr int KC = 16;       // channels per K step (one 64-byte LDS row)

// load-transform modes (template parameter of the kernel)
constexpr int MODE_PLAIN = 0, MODE_AFFINE = 1, MODE_AFFINE_SILU = 2, MODE_SILU = 3, MODE_RELU = 4;

struct ConvK {
    const float* src0; const float* src1; int C0; int C1; int ld0; int ld1;
    const float* coef_a; const float* coef_s;
    const float* w; const float* bias; const float* res; int ld_res;
    float out_scale; int act_out;
    float* out; int ld_out;
    int B, H, W, Co, CoPad, KH, KW;
    int M, HW, nchunk, nsteps, steps_per_split, splits;
    float* ws;   // split-K slabs [splits][M][Co] when splits > 1
    float* stats;   // optional fused per-channel moments [M/64][Co][2]
    const float* w_hdr;   // f16x3 only: header of the packed weights, [0] = 1 / (activation scale * weight scale)
    const unsigned* in_bound;   // f16x3 only, optional: bit pattern of a float B with |src element| <= sqrt(B) (see in_scale)
    // K-split TAIL of an otherwise unsplit grid (row-reuse kernel only, see conv_tile_cfg): pixel tiles >= tail_first are
    // computed by tail_splits workgroups each (tail_sps K-steps per workgroup) that write raw partial sums to slabs of
    // tail_rows rows [tail_splits][tail_rows][Co]; blockIdx.x >= tail_first enumerates (tile, split) pairs.
    int tail_first, tail_splits, tail_sps, tail_rows;
    int cut_chunk;   // conv_wide_kernel with splits == 2: piece 0 = chunks [0, cut_chunk), piece 1 = the rest (0 = equal pieces)
    // Fused 1x1 operand (row-reuse f16x3 kernel only): out += conv1x1(x2; w2).  x2 is a raw tensor (two sources like src),
    // scaled from its element bound like in_bound; its products are accumulated FIRST, the accumulators are then rescaled
    // by the exact power of two between the two operands' scales and the 3x3 K loop continues into them.
    const float* x2_src0; const float* x2_src1; int x2_C0, x2_C1, x2_ld0, x2_ld1;
    const char* x2_w;            // packed planes of the 1x1 weights (behind their header), nullptr = no fused operand
    const float* x2_hdr;         // their header: [0] = 1 / (activation scale * weight scale)
    const unsigned* x2_bound;
};

// f16x3 on a source with no GroupNorm in front of it (raw residual stream, attention output): the caller supplies a bound
// on the tensor's elements, taken from its per-channel moments (evc_gn_coeffs_bound_f32 / evc_moments_bound_f32), and the
// whole tensor is scaled by the power of two that brings that bound into [2^6, 2^7) before the fp16 split: nothing can
// overflow (<= 2^7 * 8 = 1024), typical elements sit far above fp16's subnormals, and the exact inverse goes into the
// accumulator scale.  Wave-uniform, evaluated once per kernel.
// A bound that is not finite (the NaN pattern the bound kernels write when a moment is NaN / inf) gives a NaN scale: the
// operands, the accumulator scale and so the whole output are NaN -- a non-finite input never becomes finite numbers.
__device__ __forceinline__ float in_scale(const ConvK& p) {
    if (!p.in_bound) return 1.0f;
    const float b = sqrtf(__uint_as_float(*p.in_bound));
    if (!(b < 3.0e38f)) return __builtin_nanf("");
    return b > 0.f ? ldexpf(1.0f, 6 - ilogbf(b)) : 1.0f;
}

__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

__device__ __forceinline__ float act_fn(float v, int act) {
    if (act == EVC_ACT_SILU) return silu_f(v);
    if (act == EVC_ACT_RELU) return fmaxf(v, 0.0f);
    return v;
}

template <int MODE>
__device__ __forceinline__ float4 transform(float4 v, const float4& a, const float4& s, bool ok) {
    if (MODE == MODE_AFFINE || MODE == MODE_AFFINE_SILU) {
        v.x = v.x * a.x + s.x; v.y = v.y * a.y + s.y; v.z = v.z * a.z + s.z; v.w = v.w * a.w + s.w;
    }
    if (MODE == MODE_AFFINE_SILU || MODE == MODE_SILU) {
        v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w);
    }
    if (MODE == MODE_RELU) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
    return v;
}

// 4 k-pairs x TM x TN MFMAs on one 8-deep k group; consecutive MFMAs go to different accumulators.
template <int TM, int TN>
__device__ __forceinline__ void mfma_group(f32x16 (&acc)[TM][TN], const float4 (&a)[TM], const float4 (&b)[TN]) {
#define EVC_MFMA_E(e)                                                                                   \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)      \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].e, b[j].e, acc[i][j], 0, 0, 0);
    EVC_MFMA_E(x) EVC_MFMA_E(y) EVC_MFMA_E(z) EVC_MFMA_E(w)
#undef EVC_MFMA_E
}

// Epilogue shared by the convolution kernels.  C/D map of the 32x32 MFMA: col = lane & 31,
// row = (r&3) + 8*(r>>2) + 4*(lane>>5).  Fuses bias + residual + scale + activation and, for full tiles, the
// per-channel GroupNorm moments of the output; split-K launches write raw partial sums to their slab instead.
// Full tiles: the wave-uniform cases (partial sums / residual / moments / output activation) are template parameters -- with
// run-time branches inside the unrolled tiles the epilogue is a long chain of tiny basic blocks, each with its own waits
// (measured on the wide kernel: 33 000 -> 13 000 cycles per 256 x 192 tile).  Same expressions, same order: bitwise the
// results of the generic form.
template <int TM, int TN, int WM, bool PARTIAL, bool RES, bool STATS, bool ACT>
__device__ __forceinline__ void conv_epilogue_full(const ConvK& p, f32x16 (&acc)[TM][TN], int m0, int n0, int split,
                                                   int wm, int wn, int l31, int half, bool tail) {
    constexpr int BM = 32 * TM * WM;
    const int ws_m0 = tail ? p.tail_first * BM : 0;          // slab row 0 = this pixel
    const size_t ws_M = tail ? (size_t)p.tail_rows : (size_t)p.M;
    const float ascale = p.w_hdr ? p.w_hdr[0] / in_scale(p) : 1.0f;
    const float oscale = p.out_scale;
    const int mw = m0 + wm * 32 * TM + 4 * half;
    const int cw = n0 + wn * 32 * TN + l31;
    const bool has_res = RES && (!ACT || p.res != nullptr), has_stats = STATS && (!ACT || p.stats != nullptr);   // ACT = the generic instance
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int co = cw + j * 32;
        const float bias = (!PARTIAL && p.bias) ? p.bias[co] : 0.f;
        float st_sum = 0.f, st_sq = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = mw + i * 32;
            if (PARTIAL) {
                float* o = p.ws + ((size_t)split * ws_M + (mb - ws_m0)) * p.Co + co;
#pragma unroll
                for (int r = 0; r < 16; ++r) o[(size_t)((r & 3) + 8 * (r >> 2)) * p.Co] = acc[i][j][r] * ascale;
            } else {
                float rv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[r] = 0.f;
                if (has_res) {
                    const float* rp = p.res + (size_t)mb * p.ld_res + co;
#pragma unroll
                    for (int r = 0; r < 16; ++r) rv[r] = rp[(size_t)((r & 3) + 8 * (r >> 2)) * p.ld_res];
                }
                float* o = p.out + (size_t)mb * p.ld_out + co;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = (acc[i][j][r] * ascale + bias + rv[r]) * oscale;
                    if (ACT) v = act_fn(v, p.act_out);
                    o[(size_t)((r & 3) + 8 * (r >> 2)) * p.ld_out] = v;
                    if (STATS) { st_sum += v; st_sq += v * v; }
                }
            }
        }
        // fused GroupNorm moments: this wave holds channel `co` of a whole 32*TM-pixel run (lanes l and l^32
        // share the channel); one writer per (run, channel) => deterministic, no atomics.
        if (has_stats) {
            st_sum += __shfl_xor(st_sum, 32);
            st_sq += __shfl_xor(st_sq, 32);
            if (half == 0) {
                float* sp = p.stats + ((size_t)(m0 / (32 * TM) + wm) * p.Co + co) * 2;
                sp[0] = st_sum; sp[1] = st_sq;
            }
        }
    }
}

template <int TM, int TN, int WM = 2>
__device__ __forceinline__ void conv_epilogue(const ConvK& p, f32x16 (&acc)[TM][TN], int m0, int n0, int split,
                                              int wm, int wn, int l31, int half, bool tail = false) {
    constexpr int BM = 32 * TM * WM;       // WM = waves along the pixel dimension (2, or 4 in the 8-wave row-reuse kernel)
    constexpr int BN = 64 * TN;
    const bool partial = p.splits > 1 || tail;
    if (m0 + BM <= p.M && n0 + BN <= p.Co) {
#define EVC_EPI(PA, RE, ST, AC) conv_epilogue_full<TM, TN, WM, PA, RE, ST, AC>(p, acc, m0, n0, split, wm, wn, l31, half, tail)
        if (partial) EVC_EPI(true, false, false, false);
        else if (p.act_out != EVC_ACT_NONE) EVC_EPI(false, true, true, true);
        else if (p.res && p.stats) EVC_EPI(false, true, true, false);
        else if (p.stats) EVC_EPI(false, false, true, false);
        else if (p.res) EVC_EPI(false, true, false, false);
        else EVC_EPI(false, false, false, false);
#undef EVC_EPI
        return;
    }
    const int ws_m0 = tail ? p.tail_first * BM : 0;          // slab row 0 = this pixel
    const size_t ws_M = tail ? (size_t)p.tail_rows : (size_t)p.M;
    // f16x3: the operands were scaled by powers of two to sit in fp16's range; undo it here (exact)
    const float ascale = p.w_hdr ? p.w_hdr[0] / in_scale(p) : 1.0f;
    const int mw = m0 + wm * 32 * TM + 4 * half;
    const int cw = n0 + wn * 32 * TN + l31;
    {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int co = cw + j * 32;
            if (co >= p.Co) continue;
            const float bias = (!partial && p.bias) ? p.bias[co] : 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mw + i * 32 + (r & 3) + 8 * (r >> 2);
                    if (m >= p.M) continue;
                    float v = acc[i][j][r] * ascale;
                    if (partial) {
                        p.ws[((size_t)split * ws_M + (m - ws_m0)) * p.Co + co] = v;
                    } else {
                        v += bias;
                        if (p.res) v += p.res[(size_t)m * p.ld_res + co];
                        v *= p.out_scale;
                        p.out[(size_t)m * p.ld_out + co] = act_fn(v, p.act_out);
                    }
                }
            }
        }
    }
}

template <int TM, int TN, int MODE>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(ConvK p) {
    constexpr int BM = 64 * TM;
    constexpr int BN = 64 * TN;
    constexpr bool HAS_COEF = MODE == MODE_AFFINE || MODE == MODE_AFFINE_SILU;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const As = smem;                    // [2][BM][16]
    float* const Ws = smem + 2 * BM * KC;      // [2][BN][16]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, half = lane >> 5;

    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int split = blockIdx.z;
    const int s_begin = split * p.steps_per_split;
    const int s_end = min(p.nsteps, s_begin + p.steps_per_split);

    // ---- per-thread gather state: rows r0 = tid/4 and r0 + 64, 4-channel column k4 ----
    const int k4 = tid & 3;
    const int padH = p.KH >> 1, padW = p.KW >> 1;
    const int Ct = p.C0 + p.C1;
    unsigned off0[TM], off1[TM], okmask[TM];  // byte offsets of the output pixel in src0 / src1, tap-valid bits
    int rb[TM], a_lds[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = (tid >> 2) + 64 * i;
        const int m = m0 + row;
        const bool valid = m < p.M;
        const int mm = valid ? m : 0;
        const int b = mm / p.HW;
        const int rem = mm - b * p.HW;
        const int y = rem / p.W;
        const int x = rem - y * p.W;
        rb[i] = b;
        off0[i] = ((unsigned)mm * (unsigned)p.ld0 + 4u * k4) * 4u;
        off1[i] = ((unsigned)mm * (unsigned)p.ld1 + 4u * k4) * 4u;
        unsigned mask = 0;
        for (int ty = 0; ty < p.KH; ++ty)
            for (int tx = 0; tx < p.KW; ++tx) {
                const int yy = y + ty - padH, xx = x + tx - padW;
                const bool ok = valid && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
                mask |= (ok ? 1u : 0u) << (ty * p.KW + tx);
            }
        okmask[i] = mask;
        a_lds[i] = row * KC + 4 * (k4 ^ ((row >> 2) & 3));        // swizzled float offset inside an A buffer
    }

    // fragment read offsets (floats): chunk c = 2*kk + half sits at position c ^ ((row >> 2) & 3); tile and
    // wave row offsets are multiples of 16 rows, so the XOR term depends on the lane only.
    const int fsw = (l31 >> 2) & 3;
    const int rd0 = 4 * ((0 + half) ^ fsw), rd1 = 4 * ((2 + half) ^ fsw);
    const int a_rd = (wm * 32 * TM + l31) * KC, w_rd = (wn * 32 * TN + l31) * KC;

    // ---- "next step" cursor, advanced incrementally (chunk-major, taps inner) ----
    int c_chunk, c_ty, c_tx;
    {
        const int taps = p.KH * p.KW;
        c_chunk = s_begin / taps;
        const int tap = s_begin - c_chunk * taps;
        c_ty = tap / p.KW;
        c_tx = tap - c_ty * p.KW;
    }
    float4 areg[TM], ca[TM], cs[TM];
    bool aok[TM];

    auto load_coefs = [&]() {
        if (HAS_COEF) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const size_t co = (size_t)rb[i] * Ct + c_chunk * KC + 4 * k4;
                ca[i] = *reinterpret_cast<const float4*>(p.coef_a + co);
                cs[i] = *reinterpret_cast<const float4*>(p.coef_s + co);
            }
        }
    };
    // Issue the operand fetches of the step the cursor points at: A -> registers, W -> LDS buffer `buf` by DMA.
    auto issue_loads = [&](int buf) {
        const int c = c_chunk * KC;
        const bool first = c < p.C0;                              // wave-uniform
        const char* src = reinterpret_cast<const char*>(first ? p.src0 : p.src1);
        const int ld = first ? p.ld0 : p.ld1;
        const int tap = c_ty * p.KW + c_tx;
        // uniform byte delta of this (tap, chunk) relative to the output pixel's channel 0
        const int delta = (((c_ty - padH) * p.W + (c_tx - padW)) * ld + (first ? c : c - p.C0)) * 4;
        const unsigned safe = (unsigned)((first ? c : c - p.C0) + 4 * k4) * 4u;   // pixel 0: always legal
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            aok[i] = (okmask[i] >> tap) & 1u;
            const unsigned o = aok[i] ? (first ? off0[i] : off1[i]) + (unsigned)delta : safe;
            areg[i] = *reinterpret_cast<const float4*>(src + o);
        }
        const float* wt = p.w + ((size_t)(tap * p.nchunk + c_chunk) * p.CoPad + n0) * KC;
        float* wl = Ws + buf * BN * KC;
#pragma unroll
        for (int j = 0; j < TN; ++j)     // one wave instruction moves 16 rows (1 KiB); 4 waves x TN rounds
            __builtin_amdgcn_global_load_lds((glb_void*)(wt + (size_t)(tid + 256 * j) * 4),
                                             (lds_void*)(wl + (wave * 16 + 64 * j) * KC), 16, 0, 0);
    };
    auto advance = [&]() {
        ++c_tx;
        if (c_tx == p.KW) { c_tx = 0; ++c_ty; }
        if (c_ty == p.KH) { c_ty = 0; ++c_chunk; }
    };
    auto store_a = [&](int buf) {
        float* A = As + buf * BM * KC;
#pragma unroll
        for (int i = 0; i < TM; ++i)
            *reinterpret_cast<float4*>(A + a_lds[i]) = transform<MODE>(areg[i], ca[i], cs[i], aok[i]);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (s_begin < s_end) {
        load_coefs();
        issue_loads(0);
        store_a(0);
    }
    __syncthreads();

    for (int s = s_begin; s < s_end; ++s) {
        const int buf = (s - s_begin) & 1;
        // Always prefetch: the last iteration re-fetches its own step into the idle buffers, which keeps the
        // body free of data-dependent control flow.
        if (s + 1 < s_end) {
            const int prev_chunk = c_chunk;
            advance();
            if (HAS_COEF && c_chunk != prev_chunk) load_coefs();      // wave-uniform, once per KH*KW steps
        }
        issue_loads(buf ^ 1);

        const float* Ab = As + buf * BM * KC + a_rd;
        const float* Wb = Ws + buf * BN * KC + w_rd;
        float4 a0[TM], b0[TN], a1[TM], b1[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a0[i] = *reinterpret_cast<const float4*>(Ab + i * 32 * KC + rd0);
#pragma unroll
        for (int j = 0; j < TN; ++j) b0[j] = *reinterpret_cast<const float4*>(Wb + j * 32 * KC + rd0);
#pragma unroll
        for (int i = 0; i < TM; ++i) a1[i] = *reinterpret_cast<const float4*>(Ab + i * 32 * KC + rd1);
#pragma unroll
        for (int j = 0; j < TN; ++j) b1[j] = *reinterpret_cast<const float4*>(Wb + j * 32 * KC + rd1);
        mfma_group<TM, TN>(acc, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        store_a(buf ^ 1);            // producer work for the next step sits among the second half of the MFMAs
        mfma_group<TM, TN>(acc, a1, b1);
        __syncthreads();             // also drains the W DMA (vmcnt) before anyone reads the new buffers
    }

    conv_epilogue<TM, TN>(p, acc, m0, n0, split, wm, wn, l31, half);
}

// ---------------------------------------------------------------------------------------------------------------
// fp32 on the bf16 matrix cores ("bf16x6").  gfx950 runs v_mfma_f32_32x32x16_bf16 at 16x the rate of the f32
// MFMA.  Every fp32 value is EXACTLY the sum of three bf16 values (8 + 8 + 8 significand bits):
//     x = x1 + x2 + x3,   x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2),
// bf16 x bf16 products are exact in fp32 and the MFMA accumulates in fp32, so
//     x*y = x1y1 + (x1y2 + x2y1) + (x1y3 + x2y2 + x3y1) + O(2^-24 |xy|)
// costs 6 bf16 MFMAs per 16-deep K slice (192 cycles) instead of 8 f32 MFMAs (512 cycles).  The three dropped
// terms are below one fp32 rounding of the product.  Measured on MI355X against an fp64 reference
// (tools/split_numerics.hip, K = 1728 ... 13824): max / rms error of this scheme is equal to or slightly BELOW
// that of the v_mfma_f32_32x32x2_f32 chain (e.g. K = 3456: 1.5e-6 / 2.1e-7 vs 1.9e-6 / 2.4e-7 of max|C|), and
// identical to the 9-product version.  So this is an fp32 convolution, not a reduced-precision one.
//
// Same tiling, gather, transform and epilogue as conv_igemm_kernel.  Differences: the K-step (16 channels) is ONE
// MFMA k-depth; LDS rows are 32 bytes (16 bf16) in three planes per operand, the 16-byte half of a row XOR-ed with
// (row >> 3) & 1 (conflict-free for the ds_read_b128 lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31});
// a thread stages 8 channels of one pixel (2 float4 loads -> transform -> split -> 3 ds_write_b128); the weights
// are split at pack time and arrive by LDS-DMA.  LDS = 192 * (BM + BN) bytes (60 KB at 128 x 192).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3_bf16(const float4& v0, const float4& v1, bf16x8& p1, bf16x8& p2, bf16x8& p3) {
    const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 a = (__bf16)x[j];
        const float r = x[j] - (float)a;
        const __bf16 b = (__bf16)r;
        p1[j] = a; p2[j] = b; p3[j] = (__bf16)(r - (float)b);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// fp32 on the fp16 matrix cores ("f16x3").  v_mfma_f32_32x32x16_f16 runs at the bf16 rate.  fp16 carries 11
// significand bits, so a 2-way split  x*S = h1 + h2,  h1 = fp16(x*S), h2 = fp16(x*S - h1)  represents x to 2^-22
// relative (an fp32 rounding is 2^-24) PROVIDED h2 does not fall into fp16's subnormals: both operands are scaled by
// powers of two first -- weights by the S_w that brings max|w| into [2^14, 2^15) (per tensor, found at pack time),
// activations by S_a = 8 (this arithmetic is only used where the operand is GroupNorm-normalised, FIR-filtered or
// otherwise O(1); nothing is clamped: an element with |x * S_a| > 65504 turns into inf - inf = NaN in the output, and the
// coefficient kernel reports beforehand whether one can exist, include/evc_hip.h EVC_RANGE_*) -- and the
// accumulator is multiplied by the exact inverse 1 / (S_a S_w) in the epilogue.  Three products
//     x*y ~ h1 g1 + h1 g2 + h2 g1              (the dropped h2 g2 is <= 2^-22 |xy|)
// per 16-deep K slice instead of bf16x6's six: half the MFMAs, two LDS planes per operand instead of three, a cheaper
// split.  Measured on MI355X against fp64 (tools/split_numerics.hip, profiles/r02_split_numerics.log; SiLU(normal) x
// normal, K = 1728 / 3456 / 13824): rms error / max|C| 1.12e-7 / 1.52e-7 / 3.44e-7 -- BELOW both the f32 MFMA chain
// (1.80e-7 / 2.41e-7 / 5.38e-7) and bf16x6 (1.56e-7 / 2.08e-7 / 4.70e-7): with fp32 accumulation the error is
// dominated by the accumulator roundings, and this scheme does the fewest.  Wide-dynamic-range operands (log-normal
// scales over e^+-16) do NOT fit fp16's exponent range: raw residual-stream inputs stay on bf16x6 (host's choice,
// evc_amd/scorenet.py).
constexpr float F16_ACT_SCALE = 8.0f;

__device__ __forceinline__ void split2_f16(const float4& v0, const float4& v1, float xscale, f16x8& p1, f16x8& p2) {
    const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        // no clamp: an element beyond fp16's range becomes (inf, -inf) and NaN in the accumulator -- loud, never a silently
        // saturated value; a NaN stays a NaN.  Whether it can happen is reported by evc_gn_coeffs_bound_f32 (EVC_RANGE_*).
        const float v = x[j] * xscale;
        const _Float16 a = (_Float16)v;
        p1[j] = a; p2[j] = (_Float16)(v - (float)a);
    }
}

// The two split arithmetics share their kernels: NP = planes per operand (3: bf16x6, 2: f16x3).
template <int NP> struct Split;
template <> struct Split<3> {
    typedef bf16x8 vec;
    static constexpr int NTERM = 6;
    // cross products, smallest first: (a plane, b plane)
    static __device__ __forceinline__ constexpr int qa(int t) { return t == 0 ? 2 : (t == 1 || t == 3) ? 1 : 0; }
    static __device__ __forceinline__ constexpr int qb(int t) { return t == 2 ? 2 : (t == 1 || t == 4) ? 1 : 0; }
    static __device__ __forceinline__ f32x16 mfma(const vec& a, const vec& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ void split(const float4& v0, const float4& v1, float, vec (&pl)[3]) {
        split3_bf16(v0, v1, pl[0], pl[1], pl[2]);
    }
};
template <> struct Split<2> {
    typedef f16x8 vec;
    static constexpr int NTERM = 3;
    static __device__ __forceinline__ constexpr int qa(int t) { return t == 0 ? 1 : 0; }
    static __device__ __forceinline__ constexpr int qb(int t) { return t == 1 ? 1 : 0; }
    static __device__ __forceinline__ f32x16 mfma(const vec& a, const vec& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ void split(const float4& v0, const float4& v1, float xscale, vec (&pl)[2]) {
        split2_f16(v0, v1, xscale, pl[0], pl[1]);
    }
};

// Simple schedule (one barrier per K-step, next step's operands fetched during the current one) for any NP; serves
// f16x3 wherever the row-reuse kernel does not apply (1x1 filters, odd widths).  bf16x6 uses the software-pipelined
// conv_split_kernel below, whose register juggling is specific to three planes.
template <int NP, int TM, int TN, int MODE>
__global__ __launch_bounds__(256, 2) void conv_splitn_kernel(ConvK p) {
    typedef Split<NP> SP;
    typedef typename SP::vec vec;
    constexpr int BM = 64 * TM;
    constexpr int BN = 64 * TN;
    constexpr int RB = 32;                             // bytes per LDS row: 16 two-byte elements
    constexpr int NPIECE = NP * 2 * TN;                // weight DMA pieces of 1 KiB (32 rows of one plane) per K-step
    constexpr int NWD = (NPIECE + 3) / 4;              // ... per wave
    constexpr bool HAS_COEF = MODE == MODE_AFFINE || MODE == MODE_AFFINE_SILU;
    extern __shared__ __attribute__((aligned(16))) char smem_b[];
    char* const As = smem_b;                           // [2][NP][BM][32 B]
    char* const Ws = smem_b + 2 * NP * BM * RB;        // [2][NP][BN][32 B]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, half = lane >> 5;

    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int split = blockIdx.z;
    const int s_begin = split * p.steps_per_split;
    const int s_end = min(p.nsteps, s_begin + p.steps_per_split);

    // ---- per-thread gather state: pixel row tid/2, channel half kh (8 channels) ----
    const int row = tid >> 1, kh = tid & 1;
    const bool a_active = TM == 2 || wave < 2;                        // wave-uniform (TM = 1: rows 0..63)
    const int padH = p.KH >> 1, padW = p.KW >> 1;
    const int Ct = p.C0 + p.C1;
    unsigned off0, off1, okmask = 0;
    int rb;
    {
        const int m = m0 + row;
        const bool valid = a_active && m < p.M;
        const int mm = valid ? m : 0;
        const int b = mm / p.HW;
        const int rem = mm - b * p.HW;
        const int y = rem / p.W;
        const int x = rem - y * p.W;
        rb = b;
        off0 = ((unsigned)mm * (unsigned)p.ld0 + 8u * kh) * 4u;
        off1 = ((unsigned)mm * (unsigned)p.ld1 + 8u * kh) * 4u;
        for (int ty = 0; ty < p.KH; ++ty)
            for (int tx = 0; tx < p.KW; ++tx) {
                const int yy = y + ty - padH, xx = x + tx - padW;
                const bool ok = valid && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
                okmask |= (ok ? 1u : 0u) << (ty * p.KW + tx);
            }
    }
    const int a_lds = row * RB + 16 * (kh ^ ((row >> 3) & 1));        // byte offset inside one plane
    const int fr = l31 * RB + 16 * (half ^ ((l31 >> 3) & 1));
    const int a_rd = wm * 32 * TM * RB + fr, w_rd = wn * 32 * TN * RB + fr;

    unsigned wsrc[NWD];      // per-lane byte offset inside a (tap, chunk) slab
    int wdst[NWD];           // LDS byte offset inside a W buffer (scalar)
#pragma unroll
    for (int j = 0; j < NWD; ++j) {
        const int idx = min(wave + 4 * j, NPIECE - 1);                // surplus waves repeat the last piece (same bytes)
        const int part = idx / (2 * TN), seg = idx - part * (2 * TN);
        wsrc[j] = (unsigned)((part * p.CoPad + n0 + seg * 32) * RB + lane * 16);
        wdst[j] = (part * BN + seg * 32) * RB;
    }
    const unsigned slab = (unsigned)NP * (unsigned)p.CoPad * RB;       // bytes per (tap, chunk)
    const unsigned w_tap = (unsigned)p.nchunk * slab;                  // tap -> tap + 1
    const unsigned w_wrap = slab - (unsigned)(p.KH * p.KW) * w_tap;    // last tap of chunk c -> tap 0 of chunk c + 1

    int c_chunk, c_ty, c_tx;
    {
        const int taps = p.KH * p.KW;
        c_chunk = s_begin / taps;
        const int tap = s_begin - c_chunk * taps;
        c_ty = tap / p.KW;
        c_tx = tap - c_ty * p.KW;
    }
    unsigned w_off = (unsigned)((c_ty * p.KW + c_tx) * p.nchunk + c_chunk) * slab;
    const char* a_src; int a_px, a_delta; unsigned a_safe; bool a_first;
    auto chunk_setup = [&]() {
        const int c = c_chunk * KC;
        a_first = c < p.C0;
        a_src = reinterpret_cast<const char*>(a_first ? p.src0 : p.src1);
        a_px = (a_first ? p.ld0 : p.ld1) * 4;
        const int cc = a_first ? c : c - p.C0;
        a_delta = ((c_ty - padH) * p.W + (c_tx - padW)) * a_px + cc * 4;
        a_safe = (unsigned)(cc + 8 * kh) * 4u;
    };
    float4 areg[2] = {}, ca[2], cs[2];
    bool aok = false;
    const float xscale = F16_ACT_SCALE * in_scale(p);      // used by the fp16 split only
    auto load_coefs = [&]() {
        if (HAS_COEF) {
            const size_t co = (size_t)rb * Ct + c_chunk * KC + 8 * kh;
            ca[0] = *reinterpret_cast<const float4*>(p.coef_a + co);
            ca[1] = *reinterpret_cast<const float4*>(p.coef_a + co + 4);
            cs[0] = *reinterpret_cast<const float4*>(p.coef_s + co);
            cs[1] = *reinterpret_cast<const float4*>(p.coef_s + co + 4);
        }
    };
    auto issue_loads = [&](int buf) {
        if (a_active) {
            aok = (okmask >> (c_ty * p.KW + c_tx)) & 1u;
            const unsigned o = aok ? (a_first ? off0 : off1) + (unsigned)a_delta : a_safe;
            areg[0] = *reinterpret_cast<const float4*>(a_src + o);
            areg[1] = *reinterpret_cast<const float4*>(a_src + o + 16);
        }
        const char* wt = reinterpret_cast<const char*>(p.w) + w_off;
        char* wl = Ws + buf * NP * BN * RB;
#pragma unroll
        for (int j = 0; j < NWD; ++j)
            __builtin_amdgcn_global_load_lds((glb_void*)(wt + wsrc[j]), (lds_void*)(wl + wdst[j]), 16, 0, 0);
    };
    auto advance = [&]() -> bool {          // true when the cursor moved to a new channel chunk
        ++c_tx; a_delta += a_px; w_off += w_tap;
        if (c_tx == p.KW) { c_tx = 0; ++c_ty; a_delta += (p.W - p.KW) * a_px; }
        if (c_ty == p.KH) { c_ty = 0; ++c_chunk; w_off += w_wrap; chunk_setup(); return true; }
        return false;
    };
    auto store_a = [&](int buf) {
        if (!a_active) return;
        vec pl[NP];
        SP::split(transform<MODE>(areg[0], ca[0], cs[0], aok), transform<MODE>(areg[1], ca[1], cs[1], aok), xscale, pl);
        char* A = As + buf * NP * BM * RB + a_lds;
#pragma unroll
        for (int q = 0; q < NP; ++q) *reinterpret_cast<vec*>(A + q * BM * RB) = pl[q];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    chunk_setup();
    if (s_begin < s_end) {
        load_coefs();
        issue_loads(0);
        store_a(0);
    }
    __syncthreads();

    for (int s = s_begin; s < s_end; ++s) {
        const int buf = (s - s_begin) & 1;
        // Always prefetch: the last iteration re-fetches its own step into the idle buffers, which keeps the
        // body free of data-dependent control flow.
        if (s + 1 < s_end) {
            if (advance() && HAS_COEF) load_coefs();                  // wave-uniform, once per KH*KW steps
        }
        issue_loads(buf ^ 1);

        const char* Ab = As + buf * NP * BM * RB + a_rd;
        const char* Wb = Ws + buf * NP * BN * RB + w_rd;
        vec a[TM][NP], b[TN][NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i][q] = *reinterpret_cast<const vec*>(Ab + (q * BM + i * 32) * RB);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j][q] = *reinterpret_cast<const vec*>(Wb + (q * BN + j * 32) * RB);
        }
#pragma unroll
        for (int t = 0; t < SP::NTERM; ++t) {
            if (t == (SP::NTERM + 1) / 2) {
                __builtin_amdgcn_sched_barrier(0);
                store_a(buf ^ 1);           // producer work for the next step sits among the later MFMAs
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = SP::mfma(a[i][SP::qa(t)], b[j][SP::qb(t)], acc[i][j]);
        }
        __syncthreads();             // also drains the W DMA (vmcnt) before anyone reads the new buffers
    }

    conv_epilogue<TM, TN>(p, acc, m0, n0, split, wm, wn, l31, half);
}

// Software-pipelined schedule (default).  Iteration t of the K loop (local step index), buffers cur = t & 1,
// nxt = cur ^ 1:
//     top          read the fragments of step t that were not prefetched (a1, b1, a0, b2) from cur -- no barrier
//                  in front of them, their latency sits behind the first MFMA term (a2 x b0, prefetched)
//     terms 1-3    18 MFMAs
//     BARRIER      (a) every wave has all of step t's fragments in registers -> cur may be overwritten,
//                  (b) publishes nxt = step t+1 (filled during the previous iteration's second half)
//     second half  fill cur with step t+2: weight slab by LDS-DMA, activation tile from the registers loaded one
//                  iteration ago (transform + split + ds_write); issue the activation loads of step t+3;
//                  prefetch a2, b0 of step t+1 from nxt into the registers a2 / b2 just vacated
//     terms 4-6    18 MFMAs
// So every global load and every DMA has a full iteration to land, fragment-read latency is never exposed behind
// a barrier, and there is still exactly one barrier per K-step and two LDS buffers.
template <int TM, int TN, int MODE>
__global__ __launch_bounds__(256, 2) void conv_split_kernel(ConvK p) {
    constexpr int BM = 64 * TM;
    constexpr int BN = 64 * TN;
    constexpr int RB = 32;                       // bytes per LDS row: 16 bf16
    constexpr int NWD = (6 * TN + 3) / 4;        // weight DMA instructions per wave and K-step
    constexpr bool HAS_COEF = MODE == MODE_AFFINE || MODE == MODE_AFFINE_SILU;
    extern __shared__ __attribute__((aligned(16))) char smem_b[];
    char* const As = smem_b;                          // [2][3][BM][32 B]
    char* const Ws = smem_b + 2 * 3 * BM * RB;        // [2][3][BN][32 B]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform by construction: keep it scalar
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, half = lane >> 5;

    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int split = blockIdx.z;
    const int s_begin = split * p.steps_per_split;
    const int nst = min(p.nsteps, s_begin + p.steps_per_split) - s_begin;     // K-steps of this workgroup

    // ---- per-thread gather state: pixel row tid/2, channel half kh (8 channels) ----
    const int row = tid >> 1, kh = tid & 1;
    const bool a_active = TM == 2 || wave < 2;                        // wave-uniform (TM = 1: rows 0..63)
    const int padH = p.KH >> 1, padW = p.KW >> 1;
    const int Ct = p.C0 + p.C1;
    unsigned off0, off1, okmask = 0;
    int rb;
    {
        const int m = m0 + row;
        const bool valid = a_active && m < p.M;
        const int mm = valid ? m : 0;
        const int b = mm / p.HW;
        const int rem = mm - b * p.HW;
        const int y = rem / p.W;
        const int x = rem - y * p.W;
        rb = b;
        off0 = ((unsigned)mm * (unsigned)p.ld0 + 8u * kh) * 4u;
        off1 = ((unsigned)mm * (unsigned)p.ld1 + 8u * kh) * 4u;
        for (int ty = 0; ty < p.KH; ++ty)
            for (int tx = 0; tx < p.KW; ++tx) {
                const int yy = y + ty - padH, xx = x + tx - padW;
                const bool ok = valid && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
                okmask |= (ok ? 1u : 0u) << (ty * p.KW + tx);
            }
    }
    const int a_lds = row * RB + 16 * (kh ^ ((row >> 3) & 1));        // byte offset inside one plane
    const int fr = l31 * RB + 16 * (half ^ ((l31 >> 3) & 1));
    const int a_rd = wm * 32 * TM * RB + fr, w_rd = wn * 32 * TN * RB + fr;

    // ---- weight DMA pieces of this wave (fixed): 6*TN pieces of 1 KiB (32 rows of one plane) over 4 waves; when
    // that does not divide, the surplus waves of the last round repeat the final piece (identical bytes to the same
    // place), which keeps the K-step free of branches ----
    unsigned wsrc[NWD];
    int wdst[NWD];
#pragma unroll
    for (int j = 0; j < NWD; ++j) {
        const int idx = min(wave + 4 * j, 6 * TN - 1);
        const int part = idx / (2 * TN), seg = idx - part * (2 * TN);
        wsrc[j] = (unsigned)((part * p.CoPad + n0 + seg * 32) * RB + lane * 16);
        wdst[j] = (part * BN + seg * 32) * RB;
    }
    const unsigned slab = 3u * (unsigned)p.CoPad * RB;                 // bytes per (tap, chunk)
    const unsigned w_tap = (unsigned)p.nchunk * slab;                  // tap -> tap + 1
    const unsigned w_wrap = slab - (unsigned)(p.KH * p.KW) * w_tap;    // last tap of chunk c -> tap 0 of chunk c + 1

    // ---- two cursors over the K-steps (chunk-major, taps inner): W (weight DMA) and L (activation loads) ----
    int w_ty, w_tx, l_chunk, l_ty, l_tx;
    {
        const int taps = p.KH * p.KW;
        l_chunk = s_begin / taps;
        const int tap = s_begin - l_chunk * taps;
        l_ty = w_ty = tap / p.KW;
        l_tx = w_tx = tap - l_ty * p.KW;
    }
    unsigned w_off = (unsigned)((w_ty * p.KW + w_tx) * p.nchunk + l_chunk) * slab;
    const char* a_src; int a_px, a_delta; unsigned a_safe; bool a_first;
    auto chunk_setup = [&]() {
        const int c = l_chunk * KC;
        a_first = c < p.C0;
        a_src = reinterpret_cast<const char*>(a_first ? p.src0 : p.src1);
        a_px = (a_first ? p.ld0 : p.ld1) * 4;
        const int cc = a_first ? c : c - p.C0;
        a_delta = ((l_ty - padH) * p.W + (l_tx - padW)) * a_px + cc * 4;
        a_safe = (unsigned)(cc + 8 * kh) * 4u;
    };
    float4 areg[2] = {}, ca[2], cs[2];
    bool aok = false;
    auto load_coefs = [&]() {
        if (HAS_COEF) {
            const size_t co = (size_t)rb * Ct + l_chunk * KC + 8 * kh;
            ca[0] = *reinterpret_cast<const float4*>(p.coef_a + co);
            ca[1] = *reinterpret_cast<const float4*>(p.coef_a + co + 4);
            cs[0] = *reinterpret_cast<const float4*>(p.coef_s + co);
            cs[1] = *reinterpret_cast<const float4*>(p.coef_s + co + 4);
        }
    };
    auto load_a = [&]() {                                              // activation loads of the step L points at
        if (a_active) {
            aok = (okmask >> (l_ty * p.KW + l_tx)) & 1u;
            const unsigned o = aok ? (a_first ? off0 : off1) + (unsigned)a_delta : a_safe;
            areg[0] = *reinterpret_cast<const float4*>(a_src + o);
            areg[1] = *reinterpret_cast<const float4*>(a_src + o + 16);
        }
    };
    auto dma_w = [&](int buf) {                                        // weight slab of the step W points at
        const char* wt = reinterpret_cast<const char*>(p.w) + w_off;
        char* wl = Ws + buf * 3 * BN * RB;
#pragma unroll
        for (int j = 0; j < NWD; ++j)
            __builtin_amdgcn_global_load_lds((glb_void*)(wt + wsrc[j]), (lds_void*)(wl + wdst[j]), 16, 0, 0);
    };
    auto advance_w = [&]() {
        ++w_tx; w_off += w_tap;
        if (w_tx == p.KW) { w_tx = 0; ++w_ty; }
        if (w_ty == p.KH) { w_ty = 0; w_off += w_wrap; }
    };
    // the coefficients are (re)loaded when L enters a new chunk; the next store_a is the one that consumes the
    // data L now points at, so `ca / cs` always match the registers being stored
    auto advance_l = [&]() {
        ++l_tx; a_delta += a_px;
        if (l_tx == p.KW) { l_tx = 0; ++l_ty; a_delta += (p.W - p.KW) * a_px; }
        if (l_ty == p.KH) { l_ty = 0; ++l_chunk; chunk_setup(); load_coefs(); }
    };
    auto store_a = [&](int buf) {
        if (!a_active) return;
        bf16x8 p1, p2, p3;
        split3_bf16(transform<MODE>(areg[0], ca[0], cs[0], aok), transform<MODE>(areg[1], ca[1], cs[1], aok), p1, p2, p3);
        char* A = As + buf * 3 * BM * RB + a_lds;
        *reinterpret_cast<bf16x8*>(A) = p1;
        *reinterpret_cast<bf16x8*>(A + BM * RB) = p2;
        *reinterpret_cast<bf16x8*>(A + 2 * BM * RB) = p3;
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    bf16x8 a[TM][3], b[TN][3], bp[TN];      // bp: b0 of the NEXT step (prefetched into the space b2 vacates)
    if (nst > 0) {
        // ---- prologue: steps 0 and 1 into buffers 0 and 1, activation registers of step 2 in flight ----
        chunk_setup();
        load_coefs();
        load_a();
        dma_w(0);
        store_a(0);
        if (1 < nst) { advance_l(); advance_w(); }
        load_a();
        dma_w(1);
        store_a(1);
        if (2 < nst) { advance_l(); advance_w(); }
        load_a();
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i][2] = *reinterpret_cast<const bf16x8*>(As + a_rd + (2 * BM + i * 32) * RB);
#pragma unroll
        for (int j = 0; j < TN; ++j) bp[j] = *reinterpret_cast<const bf16x8*>(Ws + w_rd + (0 * BN + j * 32) * RB);
    }

#define EVC_SPLIT_TERM(qa, qb)                                                                          \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)      \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][qa], b[j][qb], acc[i][j], 0, 0, 0);
    for (int t = 0; t < nst; ++t) {
        const int cur = t & 1;
        const char* Ab = As + cur * 3 * BM * RB + a_rd;
        const char* Wb = Ws + cur * 3 * BN * RB + w_rd;
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j][0] = bp[j];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i][1] = *reinterpret_cast<const bf16x8*>(Ab + (1 * BM + i * 32) * RB);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j][1] = *reinterpret_cast<const bf16x8*>(Wb + (1 * BN + j * 32) * RB);
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i][0] = *reinterpret_cast<const bf16x8*>(Ab + (0 * BM + i * 32) * RB);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j][2] = *reinterpret_cast<const bf16x8*>(Wb + (2 * BN + j * 32) * RB);
        EVC_SPLIT_TERM(2, 0) EVC_SPLIT_TERM(1, 1) EVC_SPLIT_TERM(0, 2)
        // Barrier.  vmcnt counts in issue order and this wave's two YOUNGEST vector-memory operations are the
        // activation loads of step t + 2 (issued at the end of the previous second half, consumed by store_a
        // below): wait for everything older -- the weight DMA into nxt -- but leave those two in flight.
        // (__syncthreads would drain them too: a false dependency of ~one memory latency per K-step.)
        __builtin_amdgcn_sched_barrier(0);          // the 18 MFMAs above stay above (they are not memory operations)
        if (a_active) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // ---- second half (one basic block): fill `cur` with step t + 2, prefetch a2 / b0 of step t + 1 from `nxt`;
        // the staging arithmetic (transform + split) is spread over the gaps of the 18 MFMAs ----
        dma_w(cur);
        const char* An = As + (cur ^ 1) * 3 * BM * RB + a_rd;
        const char* Wn = Ws + (cur ^ 1) * 3 * BN * RB + w_rd;
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i][2] = *reinterpret_cast<const bf16x8*>(An + (2 * BM + i * 32) * RB);
#pragma unroll
        for (int j = 0; j < TN; ++j) bp[j] = *reinterpret_cast<const bf16x8*>(Wn + (0 * BN + j * 32) * RB);
        store_a(cur);
        EVC_SPLIT_TERM(1, 0) EVC_SPLIT_TERM(0, 1) EVC_SPLIT_TERM(0, 0)
#if EVC_SPLIT_INTERLEAVE
        __builtin_amdgcn_sched_group_barrier(0x010, NWD, 0);          // weight DMA
        __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);      // prefetch reads
#pragma unroll
        for (int g = 0; g < 6 * TM * TN / 2; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);        // one MFMA ...
            __builtin_amdgcn_sched_group_barrier(0x002, EVC_SPLIT_INTERLEAVE, 0);   // ... a few VALU in its shadow
        }
#endif
        // ---- cursors for the next iteration (scalar, occasionally branchy: kept out of the block above), then the
        // activation loads of step t + 3 -- the two youngest vector-memory operations at the next barrier ----
        __builtin_amdgcn_sched_barrier(0);
        if (t + 3 < nst) { advance_w(); advance_l(); }
        load_a();
    }
#undef EVC_SPLIT_TERM

    conv_epilogue<TM, TN>(p, acc, m0, n0, split, wm, wn, l31, half);
}


// ---------------------------------------------------------------------------------------------------------------
// Row-reuse form of the split-arithmetic convolutions (NP = 3: bf16x6, NP = 2: f16x3) for 3x3 filters on tiles made of whole image rows (128 % W == 0).
//
// Under a dense bf16 MFMA load the chip is power-limited (it holds ~1.98 GHz; removing stalls returns only partly as
// wall time), so what pays is LESS WORK per MFMA.  conv_split_kernel stages every activation element once per
// filter TAP: 9 gathers, 9 GroupNorm+SiLU evaluations, 9 exact bf16 splits, 9 LDS writes.  Here a macro-step is one
// (channel chunk, kernel row ty): the 128 pixels of input row y+ty-1 are staged ONCE -- each image row with a zero
// halo pixel on both sides, so the horizontal borders need no masking -- and the three horizontal taps tx = 0, 1, 2
// are three K-steps whose A fragments are read from that one image at row offsets shifted by tx (the XOR swizzle
// stays conflict-free under the shift for W >= 32; 2-way at W = 16, 3-way at W = 8).  Weights still arrive per
// tap by LDS-DMA.  Activation loads, transform + split arithmetic and LDS writes per MFMA: one third.
// LDS: 2 x NP planes x (128/W)(W+2) rows x 32 B for the activations + 2 x NP x BN x 32 B for the weights (bf16x6: 61-68 KB,
// f16x3: 41-45 KB).
// WM = 2: 128-pixel tile, 4 waves, 2 workgroups per CU.  WM = 4: 256-pixel tile, 8 waves, 1 workgroup per CU -- the
// weight slab is shared by twice the MFMAs (weight DMA per MFMA halved; it costs ~12 % at WM = 2); used for grids
// that still offer >= 2 rounds of 256-pixel tiles.
template <int NP, int WM, int TN, int MODE>
__global__ __launch_bounds__(128 * WM, WM == 2 ? 2 : 1) void conv_split_rr_kernel(ConvK p) {
    typedef Split<NP> SP;
    typedef typename SP::vec vec;
    constexpr int TM = 2;
    constexpr int BM = 64 * WM;
    constexpr int NT = 128 * WM;
    constexpr int BN = 64 * TN;
    constexpr int RB = 32;
    constexpr int NPIECE = NP * 2 * TN;                         // weight DMA pieces of 1 KiB per K-step
    constexpr int NWD = WM == 2 ? (NPIECE + 3) / 4 : TN;        // WM = 4: waves 0..2*NP-1 move TN pieces each
    constexpr bool HAS_COEF = MODE == MODE_AFFINE || MODE == MODE_AFFINE_SILU;
    extern __shared__ __attribute__((aligned(16))) char smem_b[];
    const int SR = (BM / p.W) * (p.W + 2);            // staged rows: every image row of the tile + 2 halo pixels
    const int APL = SR * RB;                          // bytes per activation plane
    char* const As = smem_b;                          // [2][NP][SR][32 B]
    char* const Ws = smem_b + 2 * NP * APL;           // [2][NP][BN][32 B]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, half = lane >> 5;

    // K-split tail (ConvK::tail_*): blockIdx.x beyond the unsplit tiles enumerates (tile, split) pairs of the last tiles
    const bool tail = WM == 2 && (int)blockIdx.x >= p.tail_first;
    const int tq = tail ? ((int)blockIdx.x - p.tail_first) / p.tail_splits : 0;
    const int m0 = (tail ? p.tail_first + tq : (int)blockIdx.x) * BM;
    const int n0 = blockIdx.y * BN;
    const int split = tail ? (int)blockIdx.x - p.tail_first - tq * p.tail_splits : (int)blockIdx.z;
    const int sps = tail ? p.tail_sps : p.steps_per_split;                      // multiple of 3 (host)
    const int s_begin = split * sps;
    const int nst = min(p.nsteps, s_begin + sps) - s_begin;                     // multiple of 3
    const int nmac = nst / 3;

    // ---- staging thread: pixel row tid/2 of the tile, channel half kh ----
    const int row = tid >> 1, kh = tid & 1;
    const int Ct = p.C0 + p.C1;
    unsigned off0, off1, okrow = 0;
    int rb;
    {
        const int m = m0 + row;
        const bool valid = m < p.M;
        const int mm = valid ? m : 0;
        const int b = mm / p.HW;
        const int rem = mm - b * p.HW;
        const int y = rem / p.W;
        rb = b;
        off0 = ((unsigned)mm * (unsigned)p.ld0 + 8u * kh) * 4u;
        off1 = ((unsigned)mm * (unsigned)p.ld1 + 8u * kh) * 4u;
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) {
            const int yy = y + ty - 1;
            okrow |= ((valid && yy >= 0 && yy < p.H) ? 1u : 0u) << ty;
        }
    }
    const int srow_p = row + 2 * (row / p.W) + 1;                     // staged row of this pixel (dx = 0)
    const int a_lds = srow_p * RB + 16 * (kh ^ ((srow_p >> 3) & 1));
    // ---- fragment read offsets: pixel m of the tile at horizontal tap tx sits in staged row m + 2*(m/W) + tx ----
    int ard[TM][3];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int ml = wm * 32 * TM + i * 32 + l31;
        const int base = ml + 2 * (ml / p.W);
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) {
            const int sr = base + tx;
            ard[i][tx] = sr * RB + 16 * (half ^ ((sr >> 3) & 1));
        }
    }
    const int w_rd = wn * 32 * TN * RB + l31 * RB + 16 * (half ^ ((l31 >> 3) & 1));

    unsigned wsrc[NWD];
    int wdst[NWD];
    const bool w_active = WM == 2 || wave < 2 * NP;                    // wave-uniform
#pragma unroll
    for (int j = 0; j < NWD; ++j) {
        const int idx = WM == 2 ? min(wave + 4 * j, NPIECE - 1) : min(wave * TN + j, NPIECE - 1);
        const int part = idx / (2 * TN), seg = idx - part * (2 * TN);
        wsrc[j] = (unsigned)((part * p.CoPad + n0 + seg * 32) * RB + lane * 16);
        wdst[j] = (part * BN + seg * 32) * RB;
    }
    const unsigned slab = (unsigned)NP * (unsigned)p.CoPad * RB;
    const unsigned w_tap = (unsigned)p.nchunk * slab;
    const unsigned w_wrap = slab - 9u * w_tap;

    // ---- cursors: W over K-steps (tap inner), L over macro-steps (chunk, ty) ----
    int l_chunk = s_begin / 9;
    int l_ty = (s_begin - l_chunk * 9) / 3;
    int w_ty = l_ty, w_tx = 0;
    unsigned w_off = (unsigned)((w_ty * 3) * p.nchunk + l_chunk) * slab;
    const char* a_src; int a_rowb, a_delta; bool a_first;
    unsigned a_safe;
    auto chunk_setup = [&]() {
        const int c = l_chunk * KC;
        a_first = c < p.C0;
        a_src = reinterpret_cast<const char*>(a_first ? p.src0 : p.src1);
        a_rowb = (a_first ? p.ld0 : p.ld1) * 4 * p.W;                 // bytes per image row
        const int cc = a_first ? c : c - p.C0;
        a_delta = (l_ty - 1) * a_rowb + cc * 4;
        a_safe = (unsigned)(cc + 8 * kh) * 4u;
    };
    float4 areg[2], ca[2], cs[2];
    bool aok = false;
    auto load_coefs = [&]() {
        if (HAS_COEF) {
            const size_t co = (size_t)rb * Ct + l_chunk * KC + 8 * kh;
            ca[0] = *reinterpret_cast<const float4*>(p.coef_a + co);
            ca[1] = *reinterpret_cast<const float4*>(p.coef_a + co + 4);
            cs[0] = *reinterpret_cast<const float4*>(p.coef_s + co);
            cs[1] = *reinterpret_cast<const float4*>(p.coef_s + co + 4);
        }
    };
    auto load_a = [&]() {
        aok = (okrow >> l_ty) & 1u;
        const unsigned o = aok ? (a_first ? off0 : off1) + (unsigned)a_delta : a_safe;
        areg[0] = *reinterpret_cast<const float4*>(a_src + o);
        areg[1] = *reinterpret_cast<const float4*>(a_src + o + 16);
    };
    auto advance_l = [&]() {                                           // next macro-step
        ++l_ty; a_delta += a_rowb;
        if (l_ty == 3) { l_ty = 0; ++l_chunk; chunk_setup(); load_coefs(); }
    };
    auto dma_w = [&](int wb) {
        const char* wt = reinterpret_cast<const char*>(p.w) + w_off;
        char* wl = Ws + wb * NP * BN * RB;
        if (w_active) {
#pragma unroll
            for (int j = 0; j < NWD; ++j)
                __builtin_amdgcn_global_load_lds((glb_void*)(wt + wsrc[j]), (lds_void*)(wl + wdst[j]), 16, 0, 0);
        }
    };
    auto advance_w = [&]() {
        ++w_tx; w_off += w_tap;
        if (w_tx == 3) { w_tx = 0; ++w_ty; }
        if (w_ty == 3) { w_ty = 0; w_off += w_wrap; }
    };
    const float xscale = F16_ACT_SCALE * in_scale(p);     // used by the fp16 split only
    auto store_a = [&](int ab) {                          // transform + split + LDS write of the loaded activation registers
        vec pl[NP];
        SP::split(transform<MODE>(areg[0], ca[0], cs[0], aok), transform<MODE>(areg[1], ca[1], cs[1], aok), xscale, pl);
        char* A = As + ab * NP * APL + a_lds;
#pragma unroll
        for (int q = 0; q < NP; ++q) *reinterpret_cast<vec*>(A + q * APL) = pl[q];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // zero both activation images once: the halo pixels stay zero for the whole kernel
    for (int o = tid * 16; o < 2 * NP * APL; o += NT * 16) *reinterpret_cast<float4*>(As + o) = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();

#define EVC_RR_TERMS(T0, T1)                                                                            \
    _Pragma("unroll") for (int t = T0; t < T1; ++t)                                                     \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)  \
            acc[i][j] = SP::mfma(a[i][SP::qa(t)], b[j][SP::qb(t)], acc[i][j]);
#define EVC_RR_FRAGS(TX)                                                                                \
    vec a[TM][NP], b[TN][NP];                                                                           \
    {                                                                                                   \
        const char* Ab = As + ab * NP * APL;                                                            \
        const char* Wb = Ws + wb * NP * BN * RB + w_rd;                                                 \
        _Pragma("unroll") for (int q = 0; q < NP; ++q) {                                                \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                              \
                a[i][q] = *reinterpret_cast<const vec*>(Ab + q * APL + ard[i][TX]);                     \
            _Pragma("unroll") for (int j = 0; j < TN; ++j)                                              \
                b[j][q] = *reinterpret_cast<const vec*>(Wb + (q * BN + j * 32) * RB);                   \
        }                                                                                               \
    }
#define EVC_RR_NEXT_W()                                                                                 \
    dma_w(wb ^ 1);                              /* weights of K-step sidx + 1 */                        \
    if (sidx + 2 < nst) advance_w();

    // ---- fused 1x1 operand (f16x3): out += conv1x1(x2).  One K-step per 16-channel chunk of x2: the tile's own pixels are
    // staged plain (scaled from x2's element bound), the fragments are read at the centre tap of the staged image, the
    // 1x1 weight slab arrives by LDS-DMA.  Double-buffered like the main loop, one barrier per chunk.  A split workgroup
    // takes its share of the chunks.  Saves the separate 1x1 launch, its output write and the re-read as a residual. ----
    if constexpr (NP == 2) {
        if (p.x2_w) {
            const int nch2 = (p.x2_C0 + p.x2_C1) / KC;
            const int nsp = tail ? p.tail_splits : p.splits;
            const int per = (nch2 + nsp - 1) / nsp;
            const int c_begin = min(nch2, split * per), c_end = min(nch2, c_begin + per);
            float s2 = 1.0f;
            {
                const float b = sqrtf(__uint_as_float(*p.x2_bound));
                if (!(b < 3.0e38f)) s2 = __builtin_nanf("");
                else if (b > 0.f) s2 = ldexpf(1.0f, 6 - ilogbf(b));
            }
            const float xscale2 = F16_ACT_SCALE * s2;
            const bool pvalid = m0 + row < p.M;
            const unsigned slab2 = (unsigned)NP * (unsigned)p.CoPad * RB;          // bytes per chunk of the 1x1 weights
            float4 xr[2];
            auto load2 = [&](int c) {
                const int cc = c * KC;
                const bool first = cc < p.x2_C0;
                const char* src = reinterpret_cast<const char*>(first ? p.x2_src0 : p.x2_src1);
                const unsigned ld = (unsigned)(first ? p.x2_ld0 : p.x2_ld1);
                const unsigned o = ((unsigned)(pvalid ? m0 + row : 0) * ld + (unsigned)(first ? cc : cc - p.x2_C0) + 8u * kh) * 4u;
                xr[0] = *reinterpret_cast<const float4*>(src + o);
                xr[1] = *reinterpret_cast<const float4*>(src + o + 16);
            };
            auto store2 = [&](int ab) {
                vec pl[NP];
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                SP::split(transform<MODE_PLAIN>(xr[0], z, z, pvalid), transform<MODE_PLAIN>(xr[1], z, z, pvalid), xscale2, pl);
                char* A = As + ab * NP * APL + a_lds;
#pragma unroll
                for (int q = 0; q < NP; ++q) *reinterpret_cast<vec*>(A + q * APL) = pl[q];
            };
            auto dma2 = [&](int wb2, int c) {
                const char* wt = p.x2_w + (unsigned)c * slab2;
                char* wl = Ws + wb2 * NP * BN * RB;
                if (w_active) {
#pragma unroll
                    for (int j = 0; j < NWD; ++j)
                        __builtin_amdgcn_global_load_lds((glb_void*)(wt + wsrc[j]), (lds_void*)(wl + wdst[j]), 16, 0, 0);
                }
            };
            if (c_begin < c_end) {
                load2(c_begin);
                dma2(0, c_begin);
                store2(0);
            }
            __syncthreads();
            for (int c = c_begin; c < c_end; ++c) {
                const int ab = (c - c_begin) & 1, wb = ab;
                if (c + 1 < c_end) { dma2(ab ^ 1, c + 1); load2(c + 1); }
                __builtin_amdgcn_sched_barrier(0);
                EVC_RR_FRAGS(1)
                EVC_RR_TERMS(0, SP::NTERM / 2)
                __builtin_amdgcn_sched_barrier(0);
                if (c + 1 < c_end) store2(ab ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                EVC_RR_TERMS(SP::NTERM / 2, SP::NTERM)
                __syncthreads();
            }
            // both operands into one accumulator: bring the 1x1 partial sums to the 3x3 operand's scale (power of two: exact)
            const float r = (p.x2_hdr[0] / s2) / (p.w_hdr[0] / in_scale(p));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] *= r;
        }
    }

    if (nmac > 0) {
        chunk_setup();
        load_coefs();
        load_a();
        dma_w(0);
        store_a(0);
        if (1 < nst) advance_w();
    }
    __syncthreads();

    int wb = 0, sidx = 0;        // weight buffer of the current K-step, K-step index inside this split

    for (int g = 0; g < nmac; ++g) {
        const int ab = g & 1;
        {   // ---- tx = 0: also start the activation loads of the next macro-step (the two youngest operations) ----
            EVC_RR_NEXT_W()
            __builtin_amdgcn_sched_barrier(0);          // the loads below must be issued AFTER the DMA (counted wait)
            if (g + 1 < nmac) advance_l();
            load_a();
            __builtin_amdgcn_sched_barrier(0);
            EVC_RR_FRAGS(0)
            EVC_RR_TERMS(0, SP::NTERM)
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // weight DMA landed, loads stay in flight
            __builtin_amdgcn_sched_barrier(0);
            wb ^= 1; ++sidx;
        }
        {   // ---- tx = 1 ----
            EVC_RR_NEXT_W()
            EVC_RR_FRAGS(1)
            EVC_RR_TERMS(0, SP::NTERM)
            __syncthreads();
            wb ^= 1; ++sidx;
        }
        {   // ---- tx = 2: stage the next macro-step's activation image between the two halves of the MFMAs ----
            EVC_RR_NEXT_W()
            __builtin_amdgcn_sched_barrier(0);
            EVC_RR_FRAGS(2)
            EVC_RR_TERMS(0, SP::NTERM / 2)
            __builtin_amdgcn_sched_barrier(0);
            store_a(ab ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            EVC_RR_TERMS(SP::NTERM / 2, SP::NTERM)
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            wb ^= 1; ++sidx;
        }
    }
#undef EVC_RR_TERMS
#undef EVC_RR_FRAGS
#undef EVC_RR_NEXT_W

    conv_epilogue<TM, TN, WM>(p, acc, m0, n0, split, wm, wn, l31, half, tail);
}

// Epilogue of conv_wide_kernel (full 256 x 192 tiles only).  A wave owns 128 pixels x 96 channels = 12 accumulator tiles of
// 32 x 32 in which a lane holds ONE channel of 16 pixels: written as they stand that is 192 four-byte stores per lane, and the
// store path of a CU is issue-bound (~5 B/clk measured: 19 us per tile, with every CU of a round storing at the same time).
// Here each tile goes through a wave-private LDS patch [32 pixels][32 channels] (row stride 36 floats) and leaves as 16-byte
// pieces: one global_store_dwordx4 per 8 pixels x 128 bytes of fully written lines, 48 stores per lane instead of 192.  Two
// patches per wave alternate, so a tile's read-back overlaps the next tile's arithmetic; a wave's LDS operations execute in
// order, so no barrier is involved.  Bias, residual, scale, activation and the per-channel moments (64-pixel runs, the
// layout of the split-K combine kernel, so that a K-split tail may mix both producers) are computed in the accumulator
// layout, where a lane's 16 values share their channel.  Split-K / tail workgroups write raw partial sums the same way.
// The wave-uniform cases (partial sums / residual / moments / output activation) are template parameters: with run-time
// branches inside the 12 unrolled tiles the epilogue was 11 000 lines of tiny basic blocks and took 33 000 cycles per tile.
template <bool PARTIAL, bool RES, bool STATS, bool ACT>
__device__ __forceinline__ void wide_epilogue_t(const ConvK& p, f32x16 (&acc)[4][3], char* lds, int m0, int n0, int split,
                                                int wave, int lane, bool tail) {
    constexpr int TM = 4, TN = 3, BM = 256, RS = 36;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, half = lane >> 5;
    const int ws_m0 = tail ? p.tail_first * BM : 0;
    const size_t ws_M = tail ? (size_t)p.tail_rows : (size_t)p.M;
    const float ascale = p.w_hdr[0] / in_scale(p);
    const float oscale = p.out_scale;
    // (the ACT instantiation is the generic one: residual / moments are then run-time conditions)
    const bool has_res = RES && (!ACT || p.res != nullptr), has_stats = STATS && (!ACT || p.stats != nullptr);
    float* const patch0 = reinterpret_cast<float*>(lds) + wave * 2 * 32 * RS;
    const int wr = lane >> 3, wc = (lane & 7) * 4;                   // read-back: pixel rows wr + 8 k, channels wc .. wc + 3
    const int mw = m0 + wm * 32 * TM;
    const int prow = 4 * half * RS + l31;                            // this lane's patch column; row (r & 3) + 8 (r >> 2) added per register
    // residual: the 4 x 16 bytes per lane of accumulator tile t + 3 are requested before tile t is processed (j-major order)
    constexpr int RD = 4;                                           // tiles of residual in flight (HBM latency >> one tile's arithmetic)
    float4 rq[RD][4];
    const auto res_load = [&](int t, float4 (&dst)[4]) {
        const int jj = t / TM, ii = t - jj * TM;
        const float* rp = p.res + (size_t)(mw + ii * 32 + wr) * p.ld_res + n0 + wn * 32 * TN + jj * 32 + wc;
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[k] = *reinterpret_cast<const float4*>(rp + (size_t)(8 * k) * p.ld_res);
    };
    if (has_res) {
#pragma unroll
        for (int t = 0; t < RD - 1; ++t) res_load(t, rq[t]);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int cj = n0 + wn * 32 * TN + j * 32;
        const float bias = (!PARTIAL && p.bias) ? p.bias[cj + l31] : 0.f;
        float4 ssum = make_float4(0.f, 0.f, 0.f, 0.f), ssq = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = mw + i * 32;
            const int tt = j * TM + i;
            if (has_res && tt + RD - 1 < TM * TN) res_load(tt + RD - 1, rq[(tt + RD - 1) % RD]);
            // accumulator layout (a lane = one channel): scale + bias, then through the patch
            float* const P = patch0 + ((i * TN + j) & 1) * 32 * RS;
#pragma unroll
            for (int r = 0; r < 16; ++r) P[prow + ((r & 3) + 8 * (r >> 2)) * RS] = acc[i][j][r] * ascale + bias;
            // read-back layout (a lane = 4 channels of pixel rows wr + 8 k): residual, output scale, activation, moments, store
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float4 q = *reinterpret_cast<const float4*>(P + (wr + 8 * k) * RS + wc);
                const int m = mb + wr + 8 * k;
                if (!PARTIAL) {
                    if (has_res) { const float4& rr = rq[tt % RD][k]; q.x += rr.x; q.y += rr.y; q.z += rr.z; q.w += rr.w; }
                    q.x *= oscale; q.y *= oscale; q.z *= oscale; q.w *= oscale;
                    if (ACT) { q.x = act_fn(q.x, p.act_out); q.y = act_fn(q.y, p.act_out); q.z = act_fn(q.z, p.act_out); q.w = act_fn(q.w, p.act_out); }
                    if (STATS) {
                        ssum.x += q.x; ssum.y += q.y; ssum.z += q.z; ssum.w += q.w;
                        ssq.x += q.x * q.x; ssq.y += q.y * q.y; ssq.z += q.z * q.z; ssq.w += q.w * q.w;
                    }
                }
                float* dst = PARTIAL ? p.ws + ((size_t)split * ws_M + (m - ws_m0)) * p.Co + cj + wc
                                     : p.out + (size_t)m * p.ld_out + cj + wc;
                *reinterpret_cast<float4*>(dst) = q;
            }
            if (has_stats && (i & 1)) {
                // one 64-pixel run is complete: the 8 lanes with this lane's `wc` (lane distances 8, 16, 32) hold its pixel rows
#pragma unroll
                for (int d = 8; d <= 32; d <<= 1) {
                    ssum.x += __shfl_xor(ssum.x, d); ssum.y += __shfl_xor(ssum.y, d); ssum.z += __shfl_xor(ssum.z, d); ssum.w += __shfl_xor(ssum.w, d);
                    ssq.x += __shfl_xor(ssq.x, d); ssq.y += __shfl_xor(ssq.y, d); ssq.z += __shfl_xor(ssq.z, d); ssq.w += __shfl_xor(ssq.w, d);
                }
                if (wr == 0) {          // lanes 0..7: {sum, sum of squares} of 4 consecutive channels = 32 contiguous bytes
                    float* sp = p.stats + ((size_t)(m0 / 64 + wm * 2 + (i >> 1)) * p.Co + cj + wc) * 2;
                    *reinterpret_cast<float4*>(sp) = make_float4(ssum.x, ssq.x, ssum.y, ssq.y);
                    *reinterpret_cast<float4*>(sp + 4) = make_float4(ssum.z, ssq.z, ssum.w, ssq.w);
                }
                ssum = make_float4(0.f, 0.f, 0.f, 0.f); ssq = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
}

__device__ __forceinline__ void wide_epilogue(const ConvK& p, f32x16 (&acc)[4][3], char* lds, int m0, int n0, int split,
                                              int wave, int lane, bool tail) {
    const bool partial = p.splits > 1 || tail;
    if (partial) wide_epilogue_t<true, false, false, false>(p, acc, lds, m0, n0, split, wave, lane, tail);
    else if (p.act_out != EVC_ACT_NONE) wide_epilogue_t<false, true, true, true>(p, acc, lds, m0, n0, split, wave, lane, tail);   // rare: generic
    else if (p.res && p.stats) wide_epilogue_t<false, true, true, false>(p, acc, lds, m0, n0, split, wave, lane, tail);
    else if (p.stats) wide_epilogue_t<false, false, true, false>(p, acc, lds, m0, n0, split, wave, lane, tail);
    else if (p.res) wide_epilogue_t<false, true, false, false>(p, acc, lds, m0, n0, split, wave, lane, tail);
    else wide_epilogue_t<false, false, false, false>(p, acc, lds, m0, n0, split, wave, lane, tail);
}

// ---------------------------------------------------------------------------------------------------------------
// Wide-tile form of the f16x3 3x3 convolution (round 4): 256 pixels x 192 channels per workgroup, ONE 4-wave workgroup per
// CU (one wave per SIMD, so each wave may use up to 512 VGPRs), wave tile 128 pixels x 96 channels = 4 x 3 accumulator tiles.
//
// What the row-reuse kernel's counters said (profiles/NOTES.md round 3): per MFMA it issues too much non-MFMA work -- 20
// fragment reads per 36 MFMAs, a 12 KB weight slab by LDS-DMA per 72 MFMAs per workgroup (125-226 cycles to ISSUE each
// 1 KB piece beside a busy matrix pipe), every input row gathered + GroupNorm/SiLU-transformed + split three times (once
// by each of the three tiles that need it).  Here:
//   * wave tile 128 x 96: 8 activation fragment reads per 36 MFMAs, and the weights do not touch the LDS at all -- a lane's B
//     fragment is 16 contiguous bytes of the packed weights ([tap][chunk][plane][co][16 fp16], halves pre-swizzled), so each
//     wave loads its own 6 fragments per K-step straight from L2 into registers (global_load_dwordx4, one fully coalesced
//     1 KB block per instruction), prefetched TWO K-steps ahead in a 3-deep register rotation: no LDS-DMA issue cost, no
//     weight barrier, no LDS bandwidth for half of the operand traffic;
//   * a macro-step is a whole 16-channel chunk: the tile's 256 / W image rows plus one halo row above and below are staged
//     ONCE (zero halo pixel left and right of each row) and all NINE taps read their A fragments from that image at
//     offsets ty * (W + 2) + tx -- an input row is gathered, transformed and split (R + 2) / R times instead of 3
//     (2x at 128 x 128, 1.5x at 64 x 64, 1.25x at 32 x 32), activation loads and LDS writes likewise;
//   * ONE workgroup barrier per chunk (324 MFMAs per wave): the next chunk's image is staged into the other LDS buffer
//     during K-steps 0..7 and published before K-step 8, whose A-fragment prefetch already reads it;
//   * A fragments are reloaded progressively (the two planes of pixel block i right after its 9 MFMAs were issued), B
//     fragments rotate through three register sets; with 192 accumulator registers that is ~400 VGPRs.
// Same operand arithmetic, same fused epilogue (bias, residual, scale, activation, moments in 64-pixel runs), same fused
// 1x1 operand, same K-split tail / split-K slabs as conv_split_rr_kernel; results differ from it by fp32 summation order
// only (taps inner instead of channel chunks inner per kernel row).
// Requires: f16x3, 3x3, Co % 192 == 0, H*W % 256 == 0, 256 % W == 0 (tiles = whole rows of ONE image).
#ifndef EVC_WIDE_VMEM_EVERY
#define EVC_WIDE_VMEM_EVERY 36  // one global load (weight fragment / staging) issued per this many MFMAs: 64 B/clk of L1 path per CU,
                                // so a burst of 6-12 loads per wave stalls the wave's MFMA issue behind the other waves' bursts
#endif
#ifndef EVC_WIDE_BDIST
#define EVC_WIDE_BDIST 1        // weight fragments are fetched this many K-steps ahead (1 or 2; three register sets either way)
#endif
#ifndef EVC_WIDE_BUFFER_LOADS
#define EVC_WIDE_BUFFER_LOADS 0
#endif
#ifndef EVC_WIDE_ABL
#define EVC_WIDE_ABL 0     // diagnostic builds of tools/conv_bench.hip only (WRONG results): 1 = no staging arithmetic, 2 = no weight loads in
                           // the loop, 4 = no fragment reads in the loop, 8 = no activation loads in the loop, 16 = no epilogue
#endif
#ifdef EVC_WIDE_STAMPS     // diagnostic builds of tools/conv_bench.hip: s_memtime at the phase boundaries of every workgroup
__device__ unsigned long long g_wide_stamps[8192 * 8];
#define EVC_STAMP(k) if (threadIdx.x == 0) g_wide_stamps[(blockIdx.x & 8191) * 8 + (k)] = __builtin_readcyclecounter();
#else
#define EVC_STAMP(k)
#endif
template <int MODE, int NU, bool X2>
__global__ __launch_bounds__(256, 1) void conv_wide_kernel(ConvK p) {
    typedef Split<2> SP;
    typedef f16x8 vec;
    constexpr int TM = 4, TN = 3, BM = 256, BN = 192, RB = 32;
    constexpr bool HAS_COEF = MODE == MODE_AFFINE || MODE == MODE_AFFINE_SILU;
    extern __shared__ __attribute__((aligned(16))) char smem_b[];
    const int W = p.W, SW = W + 2;
    const int R = BM / W;                              // image rows per tile
    const int SPX = (R + 2) * SW;                      // staged pixels: R + 2 rows of W + 2
    const int APL = SPX * RB;                          // bytes per activation plane
    // LDS image: [2 buffers][2 planes][2 channel halves][SPX pixels][16 B].  The two 8-channel halves of a pixel live in
    // separate regions, so the 16 lanes of a ds_read_b128 group (one half, 16 pixels with gaps of 8 / 4) read 16-byte pieces
    // 16 bytes apart: bank-conflict free under ANY tap shift with no XOR swizzle -- a fragment address is one per-lane base
    // plus a wave-uniform tap offset.
    const int HPL = SPX * 16;                          // bytes per channel half of a plane
    char* const As = smem_b;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, half = lane >> 5;

#ifdef EVC_WIDE_STAGGER_TEST     // diagnostic: delay the first-round workgroups by (block % 4) x ~4 us so that rounds end out of phase
    if (blockIdx.x < 256) for (int q = 0; q < (int)((blockIdx.x >> 3) & 3) * EVC_WIDE_STAGGER_TEST; ++q) __builtin_amdgcn_s_sleep(127);
#endif
    EVC_STAMP(0)
    // K-split tail (ConvK::tail_*): blockIdx.x beyond the unsplit tiles enumerates (tile, split) pairs of the last tiles
    const bool tail = (int)blockIdx.x >= p.tail_first;
    const int tq = tail ? ((int)blockIdx.x - p.tail_first) / p.tail_splits : 0;
    const int m0 = (tail ? p.tail_first + tq : (int)blockIdx.x) * BM;
    const int n0 = blockIdx.y * BN;
    const int split = tail ? (int)blockIdx.x - p.tail_first - tq * p.tail_splits : (int)blockIdx.z;
    const int sps = tail ? p.tail_sps : p.steps_per_split;                      // multiple of 9 (host)
    // unequal pieces (ConvK::cut_chunk): a grid of T < 256 tiles leaves 256 - T CUs idle; cutting every tile into a long piece
    // (dispatched first: blockIdx.z = 0) and a short one lets the idle CUs -- and the CUs that finish a short piece -- take
    // the short pieces while the long ones run
    const int c_begin = p.cut_chunk ? (split ? p.cut_chunk : 0) : min(p.nchunk, split * (sps / 9));
    const int c_end = p.cut_chunk ? (split ? p.nchunk : p.cut_chunk) : min(p.nchunk, c_begin + sps / 9);

    const int bimg = m0 / p.HW;                        // the tile lies inside ONE image (HW % 256 == 0)
    const int y0 = (m0 - bimg * p.HW) / W;
    const int Ct = p.C0 + p.C1;

    // ---- staging units of this thread: (staged row, column, 8-channel half); unit u = tid + 256 k ----
    const int kh = tid & 1;
    unsigned goff0[NU], goff1[NU];
    int slds[NU];
    bool sval[NU], sex[NU];
#pragma unroll
    for (int k = 0; k < NU; ++k) {
        int pp = (tid + 256 * k) >> 1;
        if (pp >= (R + 2) * W) pp = tid >> 1;          // surplus unit (W < 128): repeats this thread's unit 0 -- same bytes to
                                                       // the same place, which keeps the K-step free of branches
        const int sr = pp / W, x = pp - sr * W;
        sex[k] = true;
        const int y = y0 + sr - 1;
        sval[k] = sex[k] && y >= 0 && y < p.H;
        const unsigned pix = (unsigned)((bimg * p.H + (sval[k] ? y : 0)) * W + x);        // row 0 of the image: always legal
        goff0[k] = (pix * (unsigned)p.ld0 + 8u * kh) * 4u;
        goff1[k] = (pix * (unsigned)p.ld1 + 8u * kh) * 4u;
        const int spx = (sex[k] ? sr : 0) * SW + x + 1;
        slds[k] = kh * HPL + spx * 16;
    }
    // ---- A fragment offsets: output pixel ml of the tile at tap (ty, tx) sits at staged pixel (r + ty) * SW + x + tx ----
    int abase[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int ml = wm * 128 + i * 32 + l31;
        const int r = ml / W, x = ml - r * W;
        abase[i] = half * HPL + (r * SW + x) * 16;
    }
    const auto tap_off = [&](int t) { return ((t / 3) * SW + (t % 3)) * 16; };       // wave-uniform
    // ---- B fragment: 16 bytes per lane of the packed weights, read straight from global memory ----
    const unsigned planeB = (unsigned)p.CoPad * RB;                       // bytes per plane of a (tap, chunk) slab
    const unsigned slab = 2u * planeB;
    const unsigned w_tap = (unsigned)p.nchunk * slab;
    const unsigned wlane = (unsigned)(n0 + wn * 96 + l31) * RB + 16u * (unsigned)(half ^ ((l31 >> 3) & 1));
    const char* const wbase = reinterpret_cast<const char*>(p.w);
#if EVC_WIDE_BUFFER_LOADS
    // buffer form: one wave-uniform descriptor, a 32-bit per-lane offset and a scalar offset per (tap, chunk)
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wbase), 0, 0x7fffffff, 0x00020000);
    auto load_b = [&](vec (&bb)[TN][2], int c, int t) {
        const int so = (int)((unsigned)t * w_tap + (unsigned)c * slab);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const i32x4 q0 = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (int)wlane + j * 32 * RB, so, 0);
            const i32x4 q1 = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (int)(wlane + planeB) + j * 32 * RB, so, 0);
            bb[j][0] = __builtin_bit_cast(vec, q0);
            bb[j][1] = __builtin_bit_cast(vec, q1);
        }
    };
#else
    auto load_b = [&](vec (&bb)[TN][2], int c, int t) {
        const char* wt = wbase + ((unsigned)t * w_tap + (unsigned)c * slab);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            bb[j][0] = *reinterpret_cast<const vec*>(wt + wlane + j * 32 * RB);
            bb[j][1] = *reinterpret_cast<const vec*>(wt + planeB + wlane + j * 32 * RB);
        }
    };
#endif

    const float xscale = F16_ACT_SCALE * in_scale(p);
    float4 xr[NU][2], ca[2], cs[2];
    vec a[TM][2];
    auto stage_load = [&](int k, int c) {               // activation registers of staging unit k, chunk c
        const int cc = c * KC;
        const bool first = cc < p.C0;                   // wave-uniform
        const char* src = reinterpret_cast<const char*>(first ? p.src0 : p.src1);
        const unsigned o = (first ? goff0[k] : goff1[k]) + (unsigned)(first ? cc : cc - p.C0) * 4u;
        xr[k][0] = *reinterpret_cast<const float4*>(src + o);
        xr[k][1] = *reinterpret_cast<const float4*>(src + o + 16);
    };
    auto coef_load = [&](int c) {                       // GroupNorm coefficients of chunk c (this thread's 8 channels)
        if (HAS_COEF) {
            const size_t co = (size_t)bimg * Ct + c * KC + 8 * kh;
            ca[0] = *reinterpret_cast<const float4*>(p.coef_a + co);
            ca[1] = *reinterpret_cast<const float4*>(p.coef_a + co + 4);
            cs[0] = *reinterpret_cast<const float4*>(p.coef_s + co);
            cs[1] = *reinterpret_cast<const float4*>(p.coef_s + co + 4);
        }
    };
    auto stage_store = [&](int k, int buf) {            // transform + split + LDS write of staging unit k
        vec pl[2];
        SP::split(transform<MODE>(xr[k][0], ca[0], cs[0], sval[k]), transform<MODE>(xr[k][1], ca[1], cs[1], sval[k]), xscale, pl);
        char* A = As + buf * 2 * APL + slds[k];
        *reinterpret_cast<vec*>(A) = pl[0];
        *reinterpret_cast<vec*>(A + APL) = pl[1];
    };
    auto read_a = [&](int i, const char* img, int toff) {       // both planes of pixel block i at a tap offset
        a[i][0] = *reinterpret_cast<const vec*>(img + abase[i] + toff);
        a[i][1] = *reinterpret_cast<const vec*>(img + APL + abase[i] + toff);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // the first chunk's operands are requested before anything waits (halo zeroing, barrier): their HBM latency -- every CU of
    // a round starts at the same time -- then overlaps the rest of the prologue.  (With a fused 1x1 operand that phase comes first.)
    vec b[3][TN][2];
    const bool early = !X2 && c_begin < c_end;                      // wave-uniform
    if (early) {
        coef_load(c_begin);
#pragma unroll
        for (int k = 0; k < NU; ++k) stage_load(k, c_begin);
        load_b(b[0], c_begin, 0);
    }
    EVC_STAMP(5)
    // zero the halo columns of the activation images once (left / right of every staged row, both channel halves, both
    // planes): nothing else ever writes them; every other pixel is rewritten by each chunk's staging
    {
        const int nimg = X2 ? 3 : 2;
        const int per_img = (R + 2) * 2 * 4;                         // rows x {left, right} x {plane, half}
        for (int u = tid; u < nimg * per_img; u += 256) {
            const int img = u / per_img, q = u - img * per_img;
            const int ph = q & 3, side = (q >> 2) & 1, row = q >> 3;
            const int spx = row * SW + (side ? W + 1 : 0);
            *reinterpret_cast<float4*>(As + img * 2 * APL + (ph >> 1) * APL + (ph & 1) * HPL + spx * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __syncthreads();

#define EVC_PIN(x) asm volatile("" : "+v"(x))
#define EVC_PIN4(f) asm volatile("" : "+v"((f).x), "+v"((f).y), "+v"((f).z), "+v"((f).w))
#define EVC_WIDE_MFMA_I(I, BB)                                                                          \
    _Pragma("unroll") for (int t_ = 0; t_ < SP::NTERM; ++t_)                                            \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                  \
            acc[I][j] = SP::mfma(a[I][SP::qa(t_)], BB[j][SP::qb(t_)], acc[I][j]);

    // ---- fused 1x1 operand: out += conv1x1(x2).  One K-step (36 MFMAs per wave) per 16-channel chunk of x2: the tile's own
    // pixels are staged raw (scaled from x2's element bound) at the centre of the image, fragments read at the centre tap.
    // Software pipeline over THREE LDS images so that one barrier per chunk is enough and nothing waits behind it: during the
    // MFMAs of chunk c the fragments of chunk c + 1 are read (image published by the previous barrier), chunk c + 2 is
    // transformed and written (its loads were issued two chunks earlier) and the loads of chunk c + 4 are issued. ----
    if constexpr (X2) {
        const int nch2 = (p.x2_C0 + p.x2_C1) / KC;
        const int nsp = tail ? p.tail_splits : p.splits;
        const int per = (nch2 + nsp - 1) / nsp;
        const int c2b = min(nch2, split * per), c2e = min(nch2, c2b + per);
        float s2 = 1.0f;
        {
            const float bb = sqrtf(__uint_as_float(*p.x2_bound));
            if (!(bb < 3.0e38f)) s2 = __builtin_nanf("");
            else if (bb > 0.f) s2 = ldexpf(1.0f, 6 - ilogbf(bb));
        }
        const float xscale2 = F16_ACT_SCALE * s2;
        const unsigned slab2 = 2u * planeB;
        float4 x2r[2][2][2];                  // [register set][unit][float4]: a chunk's loads are issued TWO steps before its store
        unsigned x2o0[2], x2o1[2];
        int x2l[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int pp = (tid + 256 * k) >> 1;                       // pixel of the tile
            const int r = pp / W, x = pp - r * W;
            x2o0[k] = ((unsigned)(m0 + pp) * (unsigned)p.x2_ld0 + 8u * kh) * 4u;
            x2o1[k] = ((unsigned)(m0 + pp) * (unsigned)p.x2_ld1 + 8u * kh) * 4u;
            x2l[k] = kh * HPL + ((r + 1) * SW + x + 1) * 16;
        }
        auto load2 = [&](int c, float4 (&xr2)[2][2]) {
            const int cc = c * KC;
            const bool first = cc < p.x2_C0;
            const char* src = reinterpret_cast<const char*>(first ? p.x2_src0 : p.x2_src1);
            const unsigned cb = (unsigned)(first ? cc : cc - p.x2_C0) * 4u;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const unsigned o = (first ? x2o0[k] : x2o1[k]) + cb;
                xr2[k][0] = *reinterpret_cast<const float4*>(src + o);
                xr2[k][1] = *reinterpret_cast<const float4*>(src + o + 16);
            }
        };
        auto store2 = [&](int buf, float4 (&xr2)[2][2]) {
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                vec pl[2];
                SP::split(transform<MODE_PLAIN>(xr2[k][0], z, z, true), transform<MODE_PLAIN>(xr2[k][1], z, z, true), xscale2, pl);
                char* A = As + buf * 2 * APL + x2l[k];
                *reinterpret_cast<vec*>(A) = pl[0];
                *reinterpret_cast<vec*>(A + APL) = pl[1];
            }
        };
        vec b2[2][TN][2];
        auto load_b2 = [&](vec (&bb)[TN][2], int c) {
            const char* wt = p.x2_w + (unsigned)c * slab2;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bb[j][0] = *reinterpret_cast<const vec*>(wt + wlane + j * 32 * RB);
                bb[j][1] = *reinterpret_cast<const vec*>(wt + planeB + wlane + j * 32 * RB);
            }
        };
        const int n2 = c2e - c2b;
        if (n2 > 0) {
            const auto clampc = [&](int c) { return min(c, c2e - 1); };
            load2(c2b, x2r[0]);
            load2(clampc(c2b + 1), x2r[1]);
            load_b2(b2[0], c2b);
            store2(0, x2r[0]);
            store2(1, x2r[1]);
            load2(clampc(c2b + 2), x2r[0]);
            load2(clampc(c2b + 3), x2r[1]);
            __syncthreads();
#pragma unroll
            for (int i = 0; i < TM; ++i) read_a(i, As, tap_off(4));
            // three images: chunk k of this phase lives in image k % 3 (ring index kept without a division)
            int ring = 0;
#define EVC_WIDE_X2_STEP(BCUR, BNXT, XR)                                                                    \
            {                                                                                           \
                const int rn1 = ring == 2 ? 0 : ring + 1, rn2 = rn1 == 2 ? 0 : rn1 + 1;                 \
                _Pragma("unroll") for (int j = 0; j < TN; ++j) { EVC_PIN(BCUR[j][0]); EVC_PIN(BCUR[j][1]); } \
                _Pragma("unroll") for (int k = 0; k < 2; ++k) { EVC_PIN4(XR[k][0]); EVC_PIN4(XR[k][1]); } \
                load_b2(BNXT, clampc(c + 1));                                                           \
                const char* An = As + rn1 * 2 * APL;                                                    \
                _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                        \
                    EVC_PIN(a[i][0]); EVC_PIN(a[i][1]);                                                 \
                    EVC_WIDE_MFMA_I(i, BCUR)                                                            \
                    read_a(i, An, tap_off(4));                                                          \
                }                                                                                       \
                store2(rn2, XR);                             /* chunk c + 2 */                          \
                _Pragma("unroll") for (int m = 0; m < 36; ++m) {                                        \
                    if (m == 0) __builtin_amdgcn_sched_group_barrier(0x020, 6, 0);                      \
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                  \
                    __builtin_amdgcn_sched_group_barrier(0x402, 3, 0);                                  \
                    if (m % 9 == 8) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                  \
                }                                                                                       \
                __builtin_amdgcn_sched_group_barrier(0x200, 4, 0);                                      \
                __builtin_amdgcn_sched_barrier(0);                                                      \
                load2(clampc(c + 4), XR);                    /* stored two steps from now */            \
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                         \
                __builtin_amdgcn_sched_barrier(0);                                                      \
                ring = rn1;                                                                             \
            }
            int c = c2b;
            for (; c + 1 < c2e; c += 2) {
                EVC_WIDE_X2_STEP(b2[0], b2[1], x2r[0])
                ++c;
                EVC_WIDE_X2_STEP(b2[1], b2[0], x2r[1])
                --c;
            }
            if (c < c2e) EVC_WIDE_X2_STEP(b2[0], b2[1], x2r[0])
#undef EVC_WIDE_X2_STEP
        }
        __syncthreads();
        // both operands into one accumulator: bring the 1x1 partial sums to the 3x3 operand's scale (power of two: exact)
        const float r = (p.x2_hdr[0] / s2) / (p.w_hdr[0] / in_scale(p));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] *= r;
    }

    // ---- main loop over 16-channel chunks; 9 K-steps (taps) per chunk, fully unrolled.  Register budget (256 arch VGPRs
    // beside the 192 accumulator registers): weights of K-step T + 1 are fetched during K-step T (three named sets, two
    // live), a staging unit's activation loads are issued two K-steps before its transform + LDS write (two units live). ----
    EVC_STAMP(6)
    if (c_begin < c_end) {
        if (!early) {
            coef_load(c_begin);
#pragma unroll
            for (int k = 0; k < NU; ++k) stage_load(k, c_begin);
            load_b(b[0], c_begin, 0);
        }
        if (EVC_WIDE_BDIST == 2) load_b(b[1], c_begin, 1);
#pragma unroll
        for (int k = 0; k < NU; ++k) stage_store(k, 0);
        EVC_STAMP(7)
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TM; ++i) read_a(i, As, 0);
    }
    EVC_STAMP(1)
    for (int c = c_begin; c < c_end; ++c) {
        const int ab = (c - c_begin) & 1;
        const int cn = min(c + 1, c_end - 1);           // the last chunk restages itself into the idle buffer: no branches in the body
        const char* Acur = As + ab * 2 * APL;
        const char* Anxt = As + (ab ^ 1) * 2 * APL;
        // K-step T: MFMAs on b[T % 3] and a; fetches b[(T + 1) % 3] (weights of K-step T + 1); after the 9 MFMAs of pixel
        // block i its fragments of K-step T + 1 are read; LOADK / STOREK: staging unit whose loads are issued / whose
        // transform + split + LDS write happens in this K-step (-1 = none)
        // hipcc linearises pure instructions (MFMAs, the staging arithmetic) wherever their operands are ready, across
        // sched_barrier: each K-step therefore first "pins" the values it consumes (empty volatile asm that redefines them),
        // which keeps a K-step's MFMAs and its staging unit's arithmetic inside the K-step
#define EVC_WIDE_STEP(T, LOADK, STOREK, TAIL_CODE)                                                      \
        {                                                                                               \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) { EVC_PIN(b[T % 3][j][0]); EVC_PIN(b[T % 3][j][1]); } \
            if constexpr (STOREK >= 0 && STOREK < NU) { EVC_PIN4(xr[STOREK < 0 ? 0 : STOREK][0]); EVC_PIN4(xr[STOREK < 0 ? 0 : STOREK][1]); } \
            if constexpr (STOREK == 0 && HAS_COEF) { EVC_PIN4(ca[0]); EVC_PIN4(ca[1]); EVC_PIN4(cs[0]); EVC_PIN4(cs[1]); } \
            if (EVC_WIDE_ABL & 64) load_b(b[(T + EVC_WIDE_BDIST) % 3], c_begin, 0);   /* diagnostic: always the same (L1-resident) slab */ \
            else if (!(EVC_WIDE_ABL & 2)) load_b(b[(T + EVC_WIDE_BDIST) % 3], (T + EVC_WIDE_BDIST < 9) ? c : cn, (T + EVC_WIDE_BDIST) % 9); \
            if (T == 0 && !(EVC_WIDE_ABL & 8)) coef_load(cn);                                           \
            if constexpr (LOADK >= 0 && LOADK < NU && !(EVC_WIDE_ABL & 8)) stage_load(LOADK < 0 ? 0 : LOADK, cn); \
            const char* An = (T == 8) ? Anxt : Acur;                /* image holding K-step T + 1 */    \
            const int toff = tap_off((T + 1) % 9);                                                      \
            _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                            \
                EVC_PIN(a[i][0]); EVC_PIN(a[i][1]);                                                     \
                EVC_WIDE_MFMA_I(i, b[T % 3])                                                            \
                if (!(EVC_WIDE_ABL & 4)) read_a(i, An, toff);                                           \
            }                                                                                           \
            if constexpr (STOREK >= 0 && STOREK < NU && !(EVC_WIDE_ABL & 1)) stage_store(STOREK < 0 ? 0 : STOREK, ab ^ 1); \
            /* issue order: one global load per EVC_WIDE_VMEM_EVERY MFMAs, the staging arithmetic in the MFMAs' shadow */ \
            /* (<= 4 vector instructions each), a pixel block's two fragment reads behind its 9 MFMAs                  */ \
            _Pragma("unroll") for (int m = 0; m < 36; ++m) {                                            \
                if (m % EVC_WIDE_VMEM_EVERY == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);    \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                      \
                __builtin_amdgcn_sched_group_barrier(0x402, 4, 0);                                      \
                if (m % 9 == 8) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                      \
            }                                                                                           \
            __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);                                          \
            __builtin_amdgcn_sched_barrier(0);                                                          \
            TAIL_CODE                                                                                   \
            __builtin_amdgcn_sched_barrier(0);                                                          \
        }
        // (the GroupNorm coefficients of the chunk being staged are loaded in K-step 0: the previous chunk's were last used in
        // the previous iteration's K-step 5)
        EVC_WIDE_STEP(0, 0, -1, )
        EVC_WIDE_STEP(1, 1, -1, )
        EVC_WIDE_STEP(2, 2, 0, )
        EVC_WIDE_STEP(3, 3, 1, )
        EVC_WIDE_STEP(4, -1, 2, )
        EVC_WIDE_STEP(5, -1, 3, )
        EVC_WIDE_STEP(6, -1, -1, )
        // K-step 7 ends with the chunk's only barrier: every wave has read its K-step 8 fragments from the current image
        // and written its share of the next one
        EVC_WIDE_STEP(7, -1, -1, asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");)
        EVC_WIDE_STEP(8, -1, -1, )
#undef EVC_WIDE_STEP
    }
#undef EVC_WIDE_MFMA_I
#undef EVC_PIN
#undef EVC_PIN4

    if (EVC_WIDE_ABL & 16) {        // diagnostic: keep the accumulators alive with one store per lane
        float sacc = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) sacc += acc[i][j][e];
        p.out[(size_t)m0 * p.ld_out + tid] = sacc;
        return;
    }
    EVC_STAMP(2)
    __syncthreads();                 // every wave has finished with the activation images: they become the epilogue's patches
    wide_epilogue(p, acc, As, (EVC_WIDE_ABL & 128) ? 0 : m0, n0, split, wave, lane, tail);   // (diagnostic 128: every tile stores to tile 0's lines)
#ifdef EVC_WIDE_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    EVC_STAMP(3)
    if (threadIdx.x == 0) g_wide_stamps[(blockIdx.x & 8191) * 8 + 4] = __builtin_amdgcn_s_getreg(6 | (8 << 6) | (3 << 11)) ;   // HW_ID: CU id bits
#endif
}

// out = act((sum_z ws[z] + bias + res) * scale): deterministic split-K combine.
// Block = one 64-pixel run x 64 channels, 1024 threads (thread: channel tid & 63, rows tid >> 6, +16, ...):
// 256-byte coalesced rows, 4 rows per thread so even the 8x8 layers (M = 64 B) expose enough loads in flight,
// and -- when `stats` is given -- the per-channel {sum, sumsq} of the run in the layout of the fused conv
// epilogue (one writer per (run, channel): deterministic, no atomics).
__global__ __launch_bounds__(1024) void conv_splitk_reduce_kernel(const float* __restrict__ ws, int splits, int M,
                                                                  int Co, const float* __restrict__ bias,
                                                                  const float* __restrict__ res, int ld_res,
                                                                  float scale, int act, float* __restrict__ out,
                                                                  int ld_out, float* __restrict__ stats) {
    __shared__ float red[2][16][64];
    const int l = threadIdx.x & 63;
    const int c = blockIdx.y * 64 + l;
    const int rg = threadIdx.x >> 6;
    const int m_base = blockIdx.x * 64;
    const size_t total = (size_t)M * Co;
    float s_sum = 0.f, s_sq = 0.f;
    if (c < Co) {
        const float b = bias ? bias[c] : 0.f;
        float v[4];
        size_t idx[4];
        bool ok[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int m = m_base + rg + 16 * k;
            ok[k] = m < M;
            idx[k] = (size_t)(ok[k] ? m : 0) * Co + c;
            v[k] = 0.f;
        }
        for (int z = 0; z < splits; ++z) {          // 4 independent load streams per thread
            const float* w = ws + (size_t)z * total;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] += w[idx[k]];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!ok[k]) continue;
            const int m = m_base + rg + 16 * k;
            float o = v[k] + b;
            if (res) o += res[(size_t)m * ld_res + c];
            o = act_fn(o * scale, act);
            out[(size_t)m * ld_out + c] = o;
            s_sum += o; s_sq += o * o;
        }
    }
    if (stats) {
        red[0][rg][l] = s_sum;
        red[1][rg][l] = s_sq;
        __syncthreads();
        if (rg == 0 && c < Co) {
            float a = 0.f, q = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) { a += red[0][k][l]; q += red[1][k][l]; }
            float* sp = stats + ((size_t)blockIdx.x * Co + c) * 2;
            sp[0] = a; sp[1] = q;
        }
    }
}

// w [Co][Ci][KH][KW] -> packed [KH*KW][Ci/16][CoPad][16] (zero rows for co >= Co), 16-byte chunks swizzled.
__global__ void conv_pack_weights_kernel(const float* w, float* packed, int Co, int CoPad, int Ci, int KH, int KW) {
    const int taps = KH * KW, nchunk = Ci / KC;
    const size_t total = (size_t)taps * nchunk * CoPad * KC;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % KC);
        size_t t = i / KC;
        const int co = (int)(t % CoPad); t /= CoPad;
        const int chunk = (int)(t % nchunk);
        const int tap = (int)(t / nchunk);
        // LDS swizzle baked into the packed layout: physical 16-byte chunk q of row co holds logical chunk
        // q ^ ((co >> 2) & 3), so the kernel's linear LDS-DMA copy lands conflict-free.
        const int kl = (((k >> 2) ^ ((co >> 2) & 3)) << 2) | (k & 3);
        float v = 0.f;
        if (co < Co) v = w[((size_t)co * Ci + chunk * KC + kl) * taps + tap];
        packed[i] = v;
    }
}

// bf16x6 weights: w [Co][Ci][KH][KW] -> [KH*KW][Ci/16][3 planes][CoPad][16 bf16], 16-byte halves swizzled like the
// LDS image (half ^ ((co >> 3) & 1)) so the linear LDS-DMA copy lands conflict-free.
__global__ void conv_pack_split_kernel(const float* w, __bf16* packed, int Co, int CoPad, int Ci, int KH, int KW) {
    const int taps = KH * KW, nchunk = Ci / KC;
    const size_t total = (size_t)taps * nchunk * CoPad * KC;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % KC);
        size_t t = i / KC;
        const int co = (int)(t % CoPad); t /= CoPad;
        const int chunk = (int)(t % nchunk);
        const int tap = (int)(t / nchunk);
        float v = 0.f;
        if (co < Co) v = w[((size_t)co * Ci + chunk * KC + k) * taps + tap];
        const __bf16 a = (__bf16)v;
        const float r = v - (float)a;
        const __bf16 b = (__bf16)r;
        const __bf16 c = (__bf16)(r - (float)b);
        const int pos = (((k >> 3) ^ ((co >> 3) & 1)) << 3) | (k & 7);
        const size_t base = (size_t)(tap * nchunk + chunk) * 3 * CoPad * KC;
        packed[base + ((size_t)0 * CoPad + co) * KC + pos] = a;
        packed[base + ((size_t)1 * CoPad + co) * KC + pos] = b;
        packed[base + ((size_t)2 * CoPad + co) * KC + pos] = c;
    }
}

// f16x3 weights.  Header (F16_HDR_BYTES, written by these kernels): [0] = 1 / (S_a * S_w) as float, [1] = S_w as float,
// [2] = bit pattern of max|w| (scratch of the reduction).  Body: [KH*KW][Ci/16][2 planes][CoPad][16 fp16], halves swizzled
// like the LDS image.
constexpr int F16_HDR_BYTES = 256;

__global__ void conv_absmax_kernel(const float* w, size_t n, unsigned* out_bits) {
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        m = fmaxf(m, fabsf(w[i]));
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(out_bits, __float_as_uint(m));   // non-negative floats order like their bits
}

__device__ __forceinline__ float f16_weight_scale(unsigned max_bits) {
    const float mx = __uint_as_float(max_bits);
    if (!(mx > 0.f) || !isfinite(mx)) return 1.0f;
    return ldexpf(1.0f, 14 - ilogbf(mx));            // max|w| * S_w in [2^14, 2^15)
}

__global__ void conv_pack_f16_kernel(const float* w, char* packed, int Co, int CoPad, int Ci, int KH, int KW) {
    const int taps = KH * KW, nchunk = Ci / KC;
    const size_t total = (size_t)taps * nchunk * CoPad * KC;
    float* hdr = reinterpret_cast<float*>(packed);
    const float sw = f16_weight_scale(reinterpret_cast<const unsigned*>(packed)[2]);
    _Float16* body = reinterpret_cast<_Float16*>(packed + F16_HDR_BYTES);
    if (blockIdx.x == 0 && threadIdx.x == 0) { hdr[0] = 1.0f / (F16_ACT_SCALE * sw); hdr[1] = sw; }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % KC);
        size_t t = i / KC;
        const int co = (int)(t % CoPad); t /= CoPad;
        const int chunk = (int)(t % nchunk);
        const int tap = (int)(t / nchunk);
        float v = 0.f;
        if (co < Co) v = w[((size_t)co * Ci + chunk * KC + k) * taps + tap] * sw;
        const _Float16 a = (_Float16)v;
        const _Float16 b = (_Float16)(v - (float)a);
        const int pos = (((k >> 3) ^ ((co >> 3) & 1)) << 3) | (k & 7);
        const size_t base = (size_t)(tap * nchunk + chunk) * 2 * CoPad * KC;
        body[base + ((size_t)0 * CoPad + co) * KC + pos] = a;
        body[base + ((size_t)1 * CoPad + co) * KC + pos] = b;
    }
}

int pick_tn(int CoPad) {
    if (CoPad % 192 == 0) return 3;
    if (CoPad % 128 == 0) return 2;
    return 1;
}

}  // namespace

extern "C" int evc_conv_co_pad(int Co) { return (Co + 63) / 64 * 64; }

extern "C" long long evc_conv_packed_floats(int Co, int Ci, int KH, int KW) {
    return (long long)KH * KW * (Ci / KC) * evc_conv_co_pad(Co) * KC;
}

extern "C" int evc_conv_pack_weights_f32(const float* w, float* packed, int Co, int Ci, int KH, int KW, void* stream) {
    if (!w || !packed || Co <= 0 || Ci <= 0 || Ci % KC != 0 || KH <= 0 || KW <= 0) return EVC_EINVAL;
    const long long total = evc_conv_packed_floats(Co, Ci, KH, KW);
    int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(conv_pack_weights_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, packed, Co,
                       evc_conv_co_pad(Co), Ci, KH, KW);
    return hipGetLastError() == hipSuccess ? EVC_OK : EVC_ELAUNCH;
}

extern "C" long long evc_conv_packed_bytes(int Co, int Ci, int KH, int KW, int arith) {
    const long long elems = evc_conv_packed_floats(Co, Ci, KH, KW);
    switch (arith) {
        case EVC_ARITH_F32: return elems * 4;
        case EVC_ARITH_BF16X6: return elems * 6;                      // three bf16 planes
        case EVC_ARITH_F16X3: return F16_HDR_BYTES + elems * 4;       // header + two fp16 planes
        default: return EVC_EINVAL;
    }
}

extern "C" int evc_conv_pack_weights(const float* w, void* packed, int Co, int Ci, int KH, int KW, int arith,
                                     void* stream) {
    if (arith == EVC_ARITH_F32) return evc_conv_pack_weights_f32(w, (float*)packed, Co, Ci, KH, KW, stream);
    if (arith != EVC_ARITH_BF16X6 && arith != EVC_ARITH_F16X3) return EVC_EINVAL;
    if (!w || !packed || Co <= 0 || Ci <= 0 || Ci % KC != 0 || KH <= 0 || KW <= 0) return EVC_EINVAL;
    const long long total = evc_conv_packed_floats(Co, Ci, KH, KW);
    int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    if (arith == EVC_ARITH_BF16X6) {
        hipLaunchKernelGGL(conv_pack_split_kernel, dim3(grid), dim3(256), 0, st, w, (__bf16*)packed, Co,
                           evc_conv_co_pad(Co), Ci, KH, KW);
    } else {
        if (hipMemsetAsync(packed, 0, F16_HDR_BYTES, st) != hipSuccess) return EVC_ELAUNCH;
        const size_t n = (size_t)Co * Ci * KH * KW;
        hipLaunchKernelGGL(conv_absmax_kernel, dim3(grid), dim3(256), 0, st, w, n, reinterpret_cast<unsigned*>(packed) + 2);
        hipLaunchKernelGGL(conv_pack_f16_kernel, dim3(grid), dim3(256), 0, st, w, (char*)packed, Co,
                           evc_conv_co_pad(Co), Ci, KH, KW);
    }
    return hipGetLastError() == hipSuccess ? EVC_OK : EVC_ELAUNCH;
}

static int conv_validate(const evc_conv_args* a) {
    if (!a || !a->src0 || !a->w_packed || !a->out) return EVC_EINVAL;
    if (a->C0 <= 0 || a->C0 % KC != 0 || a->C1 < 0 || a->C1 % KC != 0) return EVC_EINVAL;
    if (a->C1 > 0 && !a->src1) return EVC_EINVAL;
    if ((a->ld0 != 0 && (a->ld0 < a->C0 || (a->ld0 & 3))) || (a->ld1 != 0 && (a->ld1 < a->C1 || (a->ld1 & 3))))
        return EVC_EINVAL;
    if ((a->coef_a == nullptr) != (a->coef_s == nullptr)) return EVC_EINVAL;
    if (a->B <= 0 || a->H <= 0 || a->W <= 0 || a->Co <= 0) return EVC_EINVAL;
    if (a->KH <= 0 || a->KW <= 0 || !(a->KH & 1) || !(a->KW & 1)) return EVC_EINVAL;
    if (a->ld_out < a->Co || (a->res && a->ld_res < a->Co)) return EVC_EINVAL;
    if ((long long)a->B * a->H * a->W > 0x7fffffffLL) return EVC_EINVAL;
    if (a->arith != EVC_ARITH_F32 && a->arith != EVC_ARITH_BF16X6 && a->arith != EVC_ARITH_F16X3) return EVC_EINVAL;
    if (a->x2_w_packed) {       // fused 1x1 operand
        if (!a->x2_src0 || !a->x2_bound || a->x2_C0 <= 0 || a->x2_C0 % KC != 0 || a->x2_C1 < 0 || a->x2_C1 % KC != 0) return EVC_EINVAL;
        if (a->x2_C1 > 0 && !a->x2_src1) return EVC_EINVAL;
        if ((a->x2_ld0 != 0 && (a->x2_ld0 < a->x2_C0 || (a->x2_ld0 & 3))) || (a->x2_ld1 != 0 && (a->x2_ld1 < a->x2_C1 || (a->x2_ld1 & 3))))
            return EVC_EINVAL;
    }
    return EVC_OK;
}

// Tile configuration + split-K choice.  Measured on MI355X (tools/conv_bench.hip): a workgroup alone on a CU is
// latency-bound (~3.1 us per K-step against 1.4 us of MFMA time), two per CU take ~3.8 us per step pair, so small
// grids want ~2.5 resident workgroups per CU (640 in all).  A grid that offers >= 384 tiles is left alone, smaller
// ones split K (at least 8 K-steps per split keep prologue / epilogue amortised).  Very small grids (< 64 tiles
// of 128 pixels: the 8x8 layers) use the 64-pixel tile (TM = 1) first: half the split factor means half the slab
// write + combine traffic and no half-empty tiles (A/B: 8x8 768->768 58 -> 69 TFLOP/s; on mid-size grids such
// as 32x32 576->576 the smaller tile loses, 92 -> 84, so it is not used there).
struct TileCfg { int tm, tn, bm, bn; long long tiles; int splits; int reuse; int wide; int steps_per_split;
                 int tail_first, tail_tiles, tail_splits, tail_sps;       // K-split tail: pixel tiles >= tail_first (0 tiles = none)
                 int cut_chunk; };                                        // wide kernel, 2 UNEQUAL K pieces: [0, cut) | [cut, nchunk); 0 = equal
static int g_wide_cut = 1;     // run-time option "wide_cut": the unequal 2-way K split of conv_wide_kernel on grids of 129..255 workgroups
static int g_tail_split = EVC_CONV_TAIL;   // run-time option "tail_split"
static int g_wide_tiles = EVC_SPLIT_WIDE_TILES;   // harness A/B switch for the 256-pixel row-reuse tiles
static int g_no_reuse = 0;     // same: lets the harness A/B the row-reuse kernel inside one binary
static int g_wide_mid = 1;     // run-time option "wide_mid": conv_wide_kernel with a uniform K split on grids below one round
static int g_wide256 = 1;      // run-time option "wide256": the 256 x 192 one-workgroup-per-CU kernel (conv_wide_kernel) on grids of >= 1 round
static int g_force_tm = 0;     // tools/conv_bench.hip (same translation unit) sets this to sweep tile heights; never set in the library

// Tile height and split-K factor of the split-arithmetic kernels: 2 workgroups per CU = 512 slots.  Every candidate
// (tile height, split factor) gets an estimated time from the grid model inside split_tile_cfg() -- rounds of 512
// workgroups x (K-steps + fixed cost per workgroup) x co-residency factor -- times
//     1 / 0.75 for 64-pixel tiles of multi-tap filters            (twice the weight traffic per MFMA; measured ~0.8)
//     (1 + c * splits / K)                                        (slab write + combine read: 8 B per output element
//                                                                  and split against 2*K FLOP; c = 40 while the slabs
//                                                                  stay in L2 / MALL, 300 beyond 192 MB)
// and the smallest wins.  Checked against measured sweeps on MI355X (tools/conv_bench.hip, mode 2; bf16x6: round 1,
// DESIGN.md section 3; f16x3: profiles/r02_conv_sweep_f16x3.log and r02_conv_sweep_f16x3_after.log): the pick is the
// measured optimum or within a few percent of it for the layer shapes of the network at B = 9.
static void split_tile_cfg(const evc_conv_args* a, long long M, long long ntile, int nsteps, TileCfg& c, bool rr_ok) {
    const double K = (double)a->KH * a->KW * (a->C0 + a->C1);
    // Time model of a grid of n equal workgroups of `sps` K-steps each, in units of one workgroup's K-step when it has
    // its CU to itself:  (sps + E) per workgroup (E = prologue + epilogue), x F2 when two share a CU, in rounds of 512
    // (2 per CU); a last partial round of <= 256 workgroups runs one per CU.  bf16x6: E = 6, F2 = 1 (a workgroup's speed is
    // set by its own dependency chain -- the round-1 sweep); f16x3: E = 20, F2 = 1.5, fitted to the round-2 sweep
    // (profiles/r02_conv_sweep_f16x3.log: its K-steps are twice as fast, so the fixed costs weigh twice as much, and two
    // workgroups on a CU do slow each other down: 32x32 384->384 at 1 / 2 / 3 / 7 splits = 195 / 255 / 327 / 292 TFLOP/s).
    const bool f16 = a->arith == EVC_ARITH_F16X3 && rr_ok;       // fitted on the 3x3 row-reuse kernel; 1x1 filters keep the old fit
    const double E = f16 ? 20.0 : 6.0, F2 = f16 ? 1.5 : 1.0;
    const auto grid_time = [&](long long n, double sps) {
        const long long R = n / 512, rem = n - R * 512;
        return (sps + E) * ((double)R * F2 + (rem == 0 ? 0.0 : (rem <= 256 && f16 ? 1.0 : F2)));
    };
    double best = 1e300;
    int best_tm = 2, best_s = 1;
    c.tail_first = 0; c.tail_tiles = 0; c.tail_splits = 1; c.tail_sps = 0;
    for (int tm = 2; tm >= 1; --tm) {
        if (g_force_tm && tm != g_force_tm) continue;
        const long long tiles = ((M + 64 * tm - 1) / (64 * tm)) * ntile;
        const int smax = a->splits > 0 ? a->splits : (nsteps / 4 < 1 ? 1 : (nsteps / 4 > 32 ? 32 : nsteps / 4));
        for (int s = a->splits > 0 ? a->splits : 1; s <= smax; ++s) {
            const double sps = (double)((nsteps + s - 1) / s);
            const double slab_mb = (double)s * (double)M * a->Co * 4.0 / 1e6;
            // slab write + combine read: 8 B per output element and split against 2*K FLOP (c = 40 while the slabs stay
            // in L2 / MALL, 300 beyond 192 MB)
            const double pen = s > 1 ? 1.0 + (slab_mb > 192.0 ? 300.0 : 40.0) * s / K : 1.0;
            // 64-pixel tiles of multi-tap filters: twice the weight traffic per MFMA (measured ~0.8); a 64-pixel tile's
            // K-step is half the work of a 128-pixel one
            // (0.55 where the 128-pixel tiles would run on the row-reuse kernel, which 64-pixel tiles cannot)
            const double tmf = tm == 1 ? (a->KH * a->KW > 1 ? 0.5 / (rr_ok ? 0.55 : 0.75) : 0.5) : 1.0;
            const double t = grid_time(tiles * s, sps) * pen * tmf;
            if (t < best * 0.9999) { best = t; best_tm = tm; best_s = s; }
        }
    }
    c.tm = best_tm;
    c.bm = 64 * c.tm;
    c.tiles = ((M + c.bm - 1) / c.bm) * ntile;
    c.splits = best_s;
    // K-split TAIL (row-reuse kernel, 128-pixel tiles, caller did not fix the split): R >= 1 full rounds of unsplit tiles,
    // then the tiles of the partial round split `ts` ways so that they fill the machine once more with short jobs --
    // instead of splitting EVERY tile (slabs + combine for the whole output) or leaving a mostly idle last round.
    // Same time model (the tail's slabs only cover the tail's rows).
    if (rr_ok && g_tail_split && a->splits <= 0 && !g_force_tm && M % 128 == 0) {
        const long long tm_all = M / 128, wgs = tm_all * ntile;
        const long long R = wgs / 512;
        const long long main_m = R * 512 / ntile, tail_m = tm_all - main_m;
        const int unit = 3 * 9;                                   // a tail workgroup runs >= 3 chunks' worth of K-steps
        if (R >= 1 && tail_m > 0 && nsteps >= 2 * unit) {
            int ts = (int)(512 / (tail_m * ntile));
            if (ts > nsteps / unit) ts = nsteps / unit;
            if (ts >= 2) {
                int sps = (nsteps + ts - 1) / ts;
                sps = (sps + 2) / 3 * 3;
                ts = (nsteps + sps - 1) / sps;
                const double t_tail = grid_time(main_m * ntile, (double)nsteps) +
                                      grid_time(tail_m * ntile * ts, (double)sps) * (1.0 + 40.0 * ts / K);
                if (ts >= 2 && t_tail < best * 0.98) {
                    c.tm = 2; c.bm = 128; c.tiles = wgs; c.splits = 1;
                    c.tail_first = (int)main_m; c.tail_tiles = (int)tail_m; c.tail_splits = ts; c.tail_sps = sps;
                }
            }
        }
    }
}

static inline bool is_split_arith(int arith) { return arith == EVC_ARITH_BF16X6 || arith == EVC_ARITH_F16X3; }
static inline int arith_planes(int arith) { return arith == EVC_ARITH_F16X3 ? 2 : 3; }
constexpr long long LDS_CAP = 160 * 1024;     // gfx950: 160 KiB per CU, one workgroup may take all of it

// dynamic LDS of the row-reuse kernel: NP planes x 32 B x (2 activation images + 2 weight buffers)
static long long rr_lds_bytes(int np, int bm, int W, int bn) {
    return (long long)np * 32 * (2LL * (bm / W) * (W + 2) + 2LL * bn);
}

// dynamic LDS of the wide kernel: 2 images (3 when a fused 1x1 operand is pipelined through them) x 2 planes x
// (256 / W + 2) rows of W + 2 pixels x 32 B
static long long wide_lds_bytes(int W, bool x2) { return (x2 ? 6LL : 4LL) * (256 / W + 2) * (W + 2) * 32; }

// conv_wide_kernel: f16x3, 3x3, 192-channel output tiles, 256-pixel tiles made of whole rows of one image
static bool wide_ok(const evc_conv_args* a) {
    const bool mode_ok = a->coef_a ? a->act_in == EVC_ACT_SILU : a->act_in == EVC_ACT_NONE;     // MODE_AFFINE_SILU / MODE_PLAIN
    return g_wide256 && mode_ok && a->arith == EVC_ARITH_F16X3 && a->KH == 3 && a->KW == 3 && a->Co % 192 == 0 &&
           (a->W == 16 || a->W == 32 || a->W == 64 || a->W == 128) && ((long long)a->H * a->W) % 256 == 0 &&
           a->ld_out % 4 == 0 && (!a->res || a->ld_res % 4 == 0);      // the epilogue moves 16-byte pieces
}

static TileCfg conv_tile_cfg(const evc_conv_args* a) {
    TileCfg c;
    c.wide = 0;
    c.cut_chunk = 0;
    const long long M = (long long)a->B * a->H * a->W;
    const int CoPad = evc_conv_co_pad(a->Co);
    c.tn = pick_tn(CoPad);
    c.bn = 64 * c.tn;
    const long long ntile = CoPad / c.bn;
    const int nsteps = a->KH * a->KW * ((a->C0 + a->C1) / KC);
    c.reuse = 0;
    if (is_split_arith(a->arith)) {
        const int np = arith_planes(a->arith);
        // row-reuse kernel: 3x3 filters, 128-pixel tiles made of whole image rows (and an LDS image that fits: W >= 4)
        const bool rr_ok = !g_no_reuse && a->KH == 3 && a->KW == 3 && a->W >= 4 && 128 % a->W == 0 &&
                           rr_lds_bytes(np, 128, a->W, c.bn) <= LDS_CAP;
        split_tile_cfg(a, M, ntile, nsteps, c, rr_ok);
        if (rr_ok && c.tm == 2) c.reuse = 1;
        // 8-wave / 256-pixel form of the row-reuse kernel (one workgroup per CU = 256 slots).  Measured (B=8, 128x128:
        // exactly 2 rounds) +4-5 %; at B=9 (2.25 rounds) -5 %: the coarser tile makes the tail worse.  So: unsplit
        // grids that are a whole number of >= 2 rounds, or long enough (>= 6 rounds) for the tail not to matter.
        if (c.reuse && c.splits == 1 && !c.tail_tiles && g_wide_tiles && M % 256 == 0 && rr_lds_bytes(np, 256, a->W, c.bn) <= LDS_CAP) {
            const long long t256 = (M / 256) * ntile;
            if ((t256 >= 512 && t256 % 256 == 0) || t256 >= 6 * 256) {
                c.bm = 256;
                c.tiles = t256;
            }
        }
    } else {
        c.tail_first = 0; c.tail_tiles = 0; c.tail_splits = 1; c.tail_sps = 0;
        c.tm = (!EVC_CONV_TM1 || ((M + 127) / 128) * ntile >= 64) ? 2 : 1;
        if (g_force_tm) c.tm = g_force_tm;
        c.bm = 64 * c.tm;
        c.tiles = ((M + c.bm - 1) / c.bm) * ntile;
        long long splits = 1;
        if (a->splits > 0) splits = a->splits;
        else if (c.tiles < 384) {
            splits = (640 + c.tiles / 2) / c.tiles;
            const int max_by_steps = nsteps / 8 > 0 ? nsteps / 8 : 1;
            if (splits > max_by_steps) splits = max_by_steps;
            if (splits > 32) splits = 32;
            if (splits < 1) splits = 1;
        }
        c.splits = (int)splits;
    }
    // Wide kernel (one 256 x 192 workgroup per CU = 256 slots per round): grids that offer at least one full round.  R full
    // rounds of unsplit tiles, then the pixel tiles of the partial round split along K (whole 16-channel chunks, >= 3 per
    // workgroup) so that they fill the machine once more -- the K-split tail of the row-reuse kernel with 256 slots.
    if (wide_ok(a) && a->splits <= 0 && !g_force_tm) {
        const long long tm_all = M / 256, nt = a->Co / 192, wgs = tm_all * nt;
        if (wgs >= 256) {
            const int nchunk = (a->C0 + a->C1) / KC;
            c.wide = 1; c.reuse = 0; c.tm = 4; c.tn = 3; c.bm = 256; c.bn = 192; c.tiles = wgs; c.splits = 1;
            c.tail_first = 0; c.tail_tiles = 0; c.tail_splits = 1; c.tail_sps = 0;
            const long long Rr = wgs / 256;
            const long long main_m = Rr * 256 / nt, tail_m = tm_all - main_m;
            if (g_tail_split && tail_m > 0) {
                int ts = (int)(256 / (tail_m * nt));
                if (ts > nchunk / 3) ts = nchunk / 3;
                if (ts >= 2) {
                    const int cps = (nchunk + ts - 1) / ts;               // chunks per tail workgroup
                    ts = (nchunk + cps - 1) / cps;
                    c.tail_first = (int)main_m; c.tail_tiles = (int)tail_m; c.tail_splits = ts; c.tail_sps = cps * 9;
                }
            }
            c.steps_per_split = nsteps;
            return c;
        }
        // Grids below one round (64x64 / 32x32 layers at the benchmark batch): still one workgroup per CU, K split uniformly
        // so that tiles x splits fills (at most) one round.  Per workgroup the kernel costs ~23 000 cycles (prologue + epilogue)
        // + ~13 900 per chunk (profiles/r04_wide_stamps_v2.log); an unsplit grid on a little over half of the CUs still beats the
        // 128-pixel tiles split three ways + their combine launch (64x64 192->192: ~95 vs 112 us).  g_wide_mid = 0 disables.
        if (g_wide_mid && wgs >= 32) {
            const int nchunk = (a->C0 + a->C1) / KC;
            int best_s = 0;
            double best_t = 1e300;
            for (int sp = 1; sp <= 8 && wgs * sp <= 256 && nchunk / sp >= 4; ++sp) {
                const int cps = (nchunk + sp - 1) / sp;
                const double slab_mb = sp > 1 ? (double)sp * M * a->Co * 4.0 / 1e6 : 0.0;
                const double t = 23000.0 + 13900.0 * cps + (sp > 1 ? 2.0 * 2000.0 * (5.0 + slab_mb / 4.0) : 0.0);   // cycles at ~2 GHz; combine: ~5 us + slabs at ~4 TB/s
                if (t < best_t) { best_t = t; best_s = sp; }
            }
            // take it when a reasonable share of the machine is busy: below that the many-small-workgroups kernel wins
            if (best_s > 0 && wgs * best_s >= 128) {
                const int cps = (nchunk + best_s - 1) / best_s;
                c.wide = 1; c.reuse = 0; c.tm = 4; c.tn = 3; c.bm = 256; c.bn = 192; c.tiles = wgs;
                c.tail_first = 0; c.tail_tiles = 0; c.tail_splits = 1; c.tail_sps = 0;
                c.splits = (nchunk + cps - 1) / cps;
                c.steps_per_split = cps * 9;
                // 129..255 unsplit workgroups (64 x 64 x 192 channels at B = 9: 144): two UNEQUAL pieces per tile.  The long
                // pieces start first on `wgs` CUs; the short ones fill the 256 - wgs idle CUs in r = ceil(wgs / (256 - wgs))
                // turns.  Same cost model: long = 23 000 + 13 900 (nchunk - s), short turns = r (23 000 + 13 900 s), + combine.
                if (g_wide_cut && best_s == 1 && wgs > 128 && nchunk >= 6) {
                    const long long idle = 256 - wgs, r = (wgs + idle - 1) / idle;
                    const double slab_mb = 2.0 * M * a->Co * 4.0 / 1e6;
                    const double comb = 2.0 * 2000.0 * (5.0 + slab_mb / 4.0);
                    int cut = 0;
                    double t_cut = best_t * 0.95;                       // must beat the unsplit grid by 5 % on the model
                    for (int sh = 1; sh <= nchunk / 2; ++sh) {
                        const double tl = 23000.0 + 13900.0 * (nchunk - sh), ts = (double)r * (23000.0 + 13900.0 * sh);
                        const double t = (tl > ts ? tl : ts) + comb;
                        if (t < t_cut) { t_cut = t; cut = nchunk - sh; }
                    }
                    if (cut > 0) { c.splits = 2; c.cut_chunk = cut; c.steps_per_split = cut * 9; }
                }
                return c;
            }
        }
    }
    // splits of the row-reuse kernel cover whole (chunk, kernel row) groups: 3 taps
    const int unit = c.reuse ? a->KW : 1;
    int sps = (nsteps + c.splits - 1) / c.splits;
    sps = (sps + unit - 1) / unit * unit;
    c.steps_per_split = sps;
    c.splits = (nsteps + sps - 1) / sps;      // no empty splits
    return c;
}

extern "C" int evc_conv_set_option(const char* name, int value) {
    if (!name) return EVC_EINVAL;
    const auto is = [&](const char* s) { int i = 0; while (s[i] && s[i] == name[i]) ++i; return s[i] == 0 && name[i] == 0; };
    if (is("wide_tiles")) { g_wide_tiles = value; return EVC_OK; }
    if (is("row_reuse")) { g_no_reuse = !value; return EVC_OK; }
    if (is("tail_split")) { g_tail_split = value; return EVC_OK; }
    if (is("wide256")) { g_wide256 = value; return EVC_OK; }
    if (is("wide_mid")) { g_wide_mid = value; return EVC_OK; }
    if (is("wide_cut")) { g_wide_cut = value; return EVC_OK; }
    return EVC_EINVAL;
}

// The fused 1x1 operand exists in the f16x3 row-reuse kernel only (3x3 filters on tiles of whole image rows).
extern "C" int evc_conv_fused_1x1_supported(const evc_conv_args* a) {
    if (conv_validate(a) != EVC_OK) return 0;
    if (a->arith != EVC_ARITH_F16X3 || a->KH != 3 || a->KW != 3) return 0;
    const TileCfg c = conv_tile_cfg(a);
    return (c.reuse == 1 || c.wide) ? 1 : 0;
}

// The kernel template instance evc_conv2d_nhwc_f32 launches for these arguments, as rocprofv3 --kernel-trace prints it
// (without the namespace), e.g. "conv_wide_kernel<2, 4, false>".  Returns the length, or EVC_EINVAL.
extern "C" int evc_conv_kernel_name(const evc_conv_args* a, char* buf, int n) {
    if (!buf || n <= 0 || conv_validate(a) != EVC_OK) return EVC_EINVAL;
    int mode;
    if (a->coef_a) mode = a->act_in == EVC_ACT_SILU ? MODE_AFFINE_SILU : MODE_AFFINE;
    else mode = a->act_in == EVC_ACT_SILU ? MODE_SILU : a->act_in == EVC_ACT_RELU ? MODE_RELU : MODE_PLAIN;
    const TileCfg c = conv_tile_cfg(a);
    int len;
    if (c.wide) len = snprintf(buf, n, "conv_wide_kernel<%d, %d, %s>", mode, a->W == 128 ? 4 : 3, a->x2_w_packed ? "true" : "false");
    else if (is_split_arith(a->arith)) {
        const int np = arith_planes(a->arith);
        if (c.reuse) len = snprintf(buf, n, "conv_split_rr_kernel<%d, %d, %d, %d>", np, c.bm == 256 ? 4 : 2, c.tn, mode);
        else if (np == 3) len = snprintf(buf, n, "conv_split_kernel<%d, %d, %d>", c.tm, c.tn, mode);
        else len = snprintf(buf, n, "conv_splitn_kernel<2, %d, %d, %d>", c.tm, c.tn, mode);
    } else len = snprintf(buf, n, "conv_igemm_kernel<%d, %d, %d>", c.tm, c.tn, mode);
    return len;
}

extern "C" int evc_conv_choose_splits(const evc_conv_args* a) {
    if (conv_validate(a) != EVC_OK) return EVC_EINVAL;
    return conv_tile_cfg(a).splits;
}

extern "C" int evc_conv_stats_splits(const evc_conv_args* a) {
    if (conv_validate(a) != EVC_OK) return 0;
    const int HW = a->H * a->W;
    if (HW % 64 != 0) return 0;
    const TileCfg c = conv_tile_cfg(a);
    if (c.splits > 1 || c.wide) return HW / 64;                // split-K combine kernel / wide kernel's epilogue: 64-pixel runs
    // (a K-split tail mixes both producers: the epilogue of the unsplit tiles and the combine of the tail write the same
    // 64-pixel runs, the checks below hold for it because its tiles are full 128-pixel tiles)
    const long long M = (long long)a->B * HW;
    const int CoPad = evc_conv_co_pad(a->Co);
    if (M % c.bm != 0 || a->Co != CoPad || a->Co % c.bn != 0) return 0;
    return HW / (32 * c.tm);                                   // produced by the conv epilogue: one run per wave row
}

extern "C" long long evc_conv_workspace_bytes(const evc_conv_args* a) {
    if (conv_validate(a) != EVC_OK) return EVC_EINVAL;
    const TileCfg c = conv_tile_cfg(a);
    if (c.tail_tiles) return (long long)c.tail_splits * c.tail_tiles * c.bm * a->Co * (long long)sizeof(float);
    if (c.splits == 1) return 0;
    return (long long)c.splits * a->B * a->H * a->W * a->Co * (long long)sizeof(float);
}

// Kernels that take more than the default 64 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize raised,
// once per (kernel instantiation, device): a bit per device id, set only after the call succeeded.
static int ensure_dynamic_lds(const void* fn, unsigned long long* done_mask, size_t lds) {
    if (lds <= 64 * 1024) return EVC_OK;
    if ((long long)lds > LDS_CAP) return EVC_EUNSUPPORTED;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return EVC_ELAUNCH;
    const unsigned long long bit = 1ULL << (dev & 63);
    if (__atomic_load_n(done_mask, __ATOMIC_ACQUIRE) & bit) return EVC_OK;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_CAP) != hipSuccess) return EVC_ELAUNCH;
    __atomic_fetch_or(done_mask, bit, __ATOMIC_RELEASE);
    return EVC_OK;
}

#ifdef EVC_DEV_FAST     // development builds only (tools/): one load-transform mode instantiated, 5x faster to compile
#define EVC_MODE_SWITCH(mode, CALL) CALL(MODE_AFFINE_SILU);
#else
#define EVC_MODE_SWITCH(mode, CALL)                                   \
    switch (mode) {                                                   \
        case MODE_AFFINE: CALL(MODE_AFFINE); break;                   \
        case MODE_AFFINE_SILU: CALL(MODE_AFFINE_SILU); break;         \
        case MODE_SILU: CALL(MODE_SILU); break;                       \
        case MODE_RELU: CALL(MODE_RELU); break;                       \
        default: CALL(MODE_PLAIN); break;                             \
    }
#endif

template <int TM, int TN>
static int launch_mode(int mode, dim3 grid, size_t lds, hipStream_t st, const ConvK& k) {
#define EVC_CALL(M) hipLaunchKernelGGL((conv_igemm_kernel<TM, TN, M>), grid, dim3(256), lds, st, k)
    EVC_MODE_SWITCH(mode, EVC_CALL)
#undef EVC_CALL
    return EVC_OK;
}

// bf16x6 off the row-reuse path: the software-pipelined kernel
template <int TM, int TN>
static int launch_split3(int mode, dim3 grid, size_t lds, hipStream_t st, const ConvK& k) {
#define EVC_CALL(M) hipLaunchKernelGGL((conv_split_kernel<TM, TN, M>), grid, dim3(256), lds, st, k)
    EVC_MODE_SWITCH(mode, EVC_CALL)
#undef EVC_CALL
    return EVC_OK;
}

// f16x3 off the row-reuse path: the simple-schedule kernel
template <int TM, int TN>
static int launch_split2(int mode, dim3 grid, size_t lds, hipStream_t st, const ConvK& k) {
#define EVC_CALL(M) hipLaunchKernelGGL((conv_splitn_kernel<2, TM, TN, M>), grid, dim3(256), lds, st, k)
    EVC_MODE_SWITCH(mode, EVC_CALL)
#undef EVC_CALL
    return EVC_OK;
}

template <int NP, int WM, int TN, int MODE>
static int launch_split_rr_one(dim3 grid, size_t lds, hipStream_t st, const ConvK& k) {
    static unsigned long long attr_done = 0;
    const int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&conv_split_rr_kernel<NP, WM, TN, MODE>), &attr_done, lds);
    if (rc != EVC_OK) return rc;
    hipLaunchKernelGGL((conv_split_rr_kernel<NP, WM, TN, MODE>), grid, dim3(128 * WM), lds, st, k);
    return EVC_OK;
}
template <int NP, int WM, int TN>
static int launch_split_rr(int mode, dim3 grid, size_t lds, hipStream_t st, const ConvK& k) {
    int rc = EVC_OK;
#define EVC_CALL(M) rc = launch_split_rr_one<NP, WM, TN, M>(grid, lds, st, k)
    EVC_MODE_SWITCH(mode, EVC_CALL)
#undef EVC_CALL
    return rc;
}

template <int MODE, int NU, bool X2>
static int launch_wide_one(dim3 grid, size_t lds, hipStream_t st, const ConvK& k) {
    static unsigned long long attr_done = 0;
    const int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&conv_wide_kernel<MODE, NU, X2>), &attr_done, lds);
    if (rc != EVC_OK) return rc;
    hipLaunchKernelGGL((conv_wide_kernel<MODE, NU, X2>), grid, dim3(256), lds, st, k);
    return EVC_OK;
}
// the wide kernel is instantiated for the two load transforms the score network's 3x3 convolutions use: GroupNorm + SiLU on
// load (MODE_AFFINE_SILU) and plain (inputs activated by the FIR pass); wide_ok() sends every other mode to the row-reuse kernel
template <int MODE>
static int launch_wide_mode(int nu, bool x2, dim3 grid, size_t lds, hipStream_t st, const ConvK& k) {
    if (nu == 4) return x2 ? launch_wide_one<MODE, 4, true>(grid, lds, st, k) : launch_wide_one<MODE, 4, false>(grid, lds, st, k);
    return x2 ? launch_wide_one<MODE, 3, true>(grid, lds, st, k) : launch_wide_one<MODE, 3, false>(grid, lds, st, k);
}
static int launch_wide(int mode, int nu, bool x2, dim3 grid, size_t lds, hipStream_t st, const ConvK& k) {
#ifdef EVC_DEV_FAST
    return launch_wide_mode<MODE_AFFINE_SILU>(nu, x2, grid, lds, st, k);
#else
    if (mode == MODE_AFFINE_SILU) return launch_wide_mode<MODE_AFFINE_SILU>(nu, x2, grid, lds, st, k);
    if (mode == MODE_PLAIN) return launch_wide_mode<MODE_PLAIN>(nu, x2, grid, lds, st, k);
    return EVC_EUNSUPPORTED;
#endif
}

#define EVC_TN_SWITCH(tn, CALL) ((tn) == 3 ? CALL(3) : (tn) == 2 ? CALL(2) : CALL(1))

static int conv2d_impl(const evc_conv_args* a, float* ws, void* stream, hipEvent_t ev_start, hipEvent_t ev_conv_end);

extern "C" int evc_conv2d_nhwc_f32(const evc_conv_args* a, float* ws, void* stream) {
    return conv2d_impl(a, ws, stream, nullptr, nullptr);
}

// Measurement hook (bench.py's roofline leg): the same launch, with `ev_start` recorded on the stream immediately before
// the convolution kernel and `ev_conv_end` immediately after it -- i.e. BEFORE the split-K combine kernel, so the interval
// is the convolution kernel's own duration, the number rocprofv3 --kernel-trace reports for it.  Either may be NULL.
extern "C" int evc_conv2d_nhwc_profiled_f32(const evc_conv_args* a, float* ws, void* stream, void* ev_start,
                                            void* ev_conv_end) {
    return conv2d_impl(a, ws, stream, (hipEvent_t)ev_start, (hipEvent_t)ev_conv_end);
}

static int conv2d_impl(const evc_conv_args* a, float* ws, void* stream, hipEvent_t ev_start, hipEvent_t ev_conv_end) {
    int rc = conv_validate(a);
    if (rc != EVC_OK) return rc;
    int mode;
    if (a->coef_a) {
        if (a->act_in == EVC_ACT_SILU) mode = MODE_AFFINE_SILU;
        else if (a->act_in == EVC_ACT_NONE) mode = MODE_AFFINE;
        else return EVC_EUNSUPPORTED;
    } else {
        mode = a->act_in == EVC_ACT_SILU ? MODE_SILU : a->act_in == EVC_ACT_RELU ? MODE_RELU : MODE_PLAIN;
        if (a->act_in != EVC_ACT_NONE && a->act_in != EVC_ACT_SILU && a->act_in != EVC_ACT_RELU) return EVC_EINVAL;
    }
    ConvK k;
    k.src0 = a->src0; k.src1 = a->src1 ? a->src1 : a->src0; k.C0 = a->C0; k.C1 = a->C1;
    k.ld0 = a->ld0 > 0 ? a->ld0 : a->C0; k.ld1 = a->ld1 > 0 ? a->ld1 : (a->C1 > 0 ? a->C1 : k.ld0);
    k.coef_a = a->coef_a; k.coef_s = a->coef_s;
    k.w = a->w_packed; k.bias = a->bias; k.res = a->res; k.ld_res = a->ld_res;
    k.w_hdr = nullptr;
    k.in_bound = nullptr;
    if (a->in_bound) {
        if (a->arith != EVC_ARITH_F16X3) return EVC_EINVAL;        // only the fp16 split scales its input
        k.in_bound = a->in_bound;
    }
    if (a->arith == EVC_ARITH_F16X3) {     // [header][planes]
        k.w_hdr = a->w_packed;
        k.w = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a->w_packed) + F16_HDR_BYTES);
    }
    k.out_scale = a->out_scale; k.act_out = a->act_out; k.out = a->out; k.ld_out = a->ld_out;
    k.B = a->B; k.H = a->H; k.W = a->W; k.Co = a->Co; k.CoPad = evc_conv_co_pad(a->Co);
    k.KH = a->KH; k.KW = a->KW;
    k.M = a->B * a->H * a->W; k.HW = a->H * a->W;
    // 32-bit byte offsets inside the kernel: each source must stay below 4 GiB
    if ((long long)k.M * k.ld0 * 4 >= (1LL << 32) || (long long)k.M * k.ld1 * 4 >= (1LL << 32)) return EVC_EUNSUPPORTED;
    if (a->KH * a->KW > 32) return EVC_EUNSUPPORTED;   // tap-valid bit mask is 32 bits wide
    k.nchunk = (a->C0 + a->C1) / KC;
    k.nsteps = a->KH * a->KW * k.nchunk;
    const TileCfg cfg = conv_tile_cfg(a);
    k.splits = cfg.splits;
    k.steps_per_split = cfg.steps_per_split;
    k.ws = ws;
    if ((k.splits > 1 || cfg.tail_tiles) && !ws) return EVC_EINVAL;
    k.stats = k.splits > 1 ? nullptr : a->stats_out;   // with split-K the combine kernel writes them
    if (a->stats_out && evc_conv_stats_splits(a) == 0) return EVC_EINVAL;
    k.x2_w = nullptr; k.x2_hdr = nullptr; k.x2_bound = nullptr; k.x2_src0 = k.x2_src1 = nullptr;
    k.x2_C0 = k.x2_C1 = k.x2_ld0 = k.x2_ld1 = 0;
    if (a->x2_w_packed) {
        if (a->arith != EVC_ARITH_F16X3 || a->KH != 3 || a->KW != 3 || !(cfg.reuse == 1 || cfg.wide)) return EVC_EUNSUPPORTED;
        k.x2_src0 = a->x2_src0; k.x2_src1 = a->x2_src1 ? a->x2_src1 : a->x2_src0;
        k.x2_C0 = a->x2_C0; k.x2_C1 = a->x2_C1;
        k.x2_ld0 = a->x2_ld0 > 0 ? a->x2_ld0 : a->x2_C0;
        k.x2_ld1 = a->x2_ld1 > 0 ? a->x2_ld1 : (a->x2_C1 > 0 ? a->x2_C1 : k.x2_ld0);
        if ((long long)k.M * k.x2_ld0 * 4 >= (1LL << 32) || (long long)k.M * k.x2_ld1 * 4 >= (1LL << 32)) return EVC_EUNSUPPORTED;
        k.x2_hdr = a->x2_w_packed;
        k.x2_w = reinterpret_cast<const char*>(a->x2_w_packed) + F16_HDR_BYTES;
        k.x2_bound = a->x2_bound;
    }
    k.tail_first = 0x7fffffff; k.tail_splits = 1; k.tail_sps = 0; k.tail_rows = 0;
    k.cut_chunk = cfg.wide ? cfg.cut_chunk : 0;
    if (cfg.tail_tiles) {
        k.tail_first = cfg.tail_first; k.tail_splits = cfg.tail_splits; k.tail_sps = cfg.tail_sps;
        k.tail_rows = cfg.tail_tiles * cfg.bm;
    }

    dim3 grid((k.M + cfg.bm - 1) / cfg.bm, k.CoPad / cfg.bn, k.splits);
    if (cfg.tail_tiles) grid.x = cfg.tail_first + cfg.tail_tiles * cfg.tail_splits;
    hipStream_t st = (hipStream_t)stream;
    if (ev_start && hipEventRecord(ev_start, st) != hipSuccess) return EVC_ELAUNCH;
    if (cfg.wide) {
        rc = launch_wide(mode, a->W == 128 ? 4 : 3, a->x2_w_packed != nullptr, grid, (size_t)wide_lds_bytes(a->W, a->x2_w_packed != nullptr), st, k);
    } else if (is_split_arith(a->arith)) {
        const int np = arith_planes(a->arith);
        if (cfg.reuse) {
            const size_t lds_rr = (size_t)rr_lds_bytes(np, cfg.bm, a->W, cfg.bn);
#define EVC_RR4_3(TN) launch_split_rr<3, 4, TN>(mode, grid, lds_rr, st, k)
#define EVC_RR2_3(TN) launch_split_rr<3, 2, TN>(mode, grid, lds_rr, st, k)
#define EVC_RR4_2(TN) launch_split_rr<2, 4, TN>(mode, grid, lds_rr, st, k)
#define EVC_RR2_2(TN) launch_split_rr<2, 2, TN>(mode, grid, lds_rr, st, k)
            if (np == 3) rc = cfg.bm == 256 ? EVC_TN_SWITCH(cfg.tn, EVC_RR4_3) : EVC_TN_SWITCH(cfg.tn, EVC_RR2_3);
            else rc = cfg.bm == 256 ? EVC_TN_SWITCH(cfg.tn, EVC_RR4_2) : EVC_TN_SWITCH(cfg.tn, EVC_RR2_2);
#undef EVC_RR4_3
#undef EVC_RR2_3
#undef EVC_RR4_2
#undef EVC_RR2_2
        } else {
            const size_t lds = (size_t)2 * np * (cfg.bm + cfg.bn) * 32;
#define EVC_S3_2(TN) launch_split3<2, TN>(mode, grid, lds, st, k)
#define EVC_S3_1(TN) launch_split3<1, TN>(mode, grid, lds, st, k)
#define EVC_S2_2(TN) launch_split2<2, TN>(mode, grid, lds, st, k)
#define EVC_S2_1(TN) launch_split2<1, TN>(mode, grid, lds, st, k)
            if (np == 3) rc = cfg.tm == 2 ? EVC_TN_SWITCH(cfg.tn, EVC_S3_2) : EVC_TN_SWITCH(cfg.tn, EVC_S3_1);
            else rc = cfg.tm == 2 ? EVC_TN_SWITCH(cfg.tn, EVC_S2_2) : EVC_TN_SWITCH(cfg.tn, EVC_S2_1);
#undef EVC_S3_2
#undef EVC_S3_1
#undef EVC_S2_2
#undef EVC_S2_1
        }
    } else {
        const size_t lds = (size_t)2 * (cfg.bm + cfg.bn) * KC * sizeof(float);
#define EVC_F_2(TN) launch_mode<2, TN>(mode, grid, lds, st, k)
#define EVC_F_1(TN) launch_mode<1, TN>(mode, grid, lds, st, k)
        rc = cfg.tm == 2 ? EVC_TN_SWITCH(cfg.tn, EVC_F_2) : EVC_TN_SWITCH(cfg.tn, EVC_F_1);
#undef EVC_F_2
#undef EVC_F_1
    }
    if (rc != EVC_OK) return rc;
    if (hipGetLastError() != hipSuccess) return EVC_ELAUNCH;
    if (ev_conv_end && hipEventRecord(ev_conv_end, st) != hipSuccess) return EVC_ELAUNCH;
    if (k.splits > 1) {
        hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3((k.M + 63) / 64, (k.Co + 63) / 64), dim3(1024), 0, st, ws,
                           k.splits, k.M, k.Co, a->bias, a->res, a->ld_res, a->out_scale, a->act_out, a->out,
                           a->ld_out, a->stats_out);
        if (hipGetLastError() != hipSuccess) return EVC_ELAUNCH;
    }
    if (cfg.tail_tiles) {       // combine of the tail's rows only
        const size_t m0 = (size_t)cfg.tail_first * cfg.bm;
        hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3((k.tail_rows + 63) / 64, (k.Co + 63) / 64), dim3(1024), 0, st, ws,
                           k.tail_splits, k.tail_rows, k.Co, a->bias, a->res ? a->res + m0 * a->ld_res : nullptr, a->ld_res,
                           a->out_scale, a->act_out, a->out + m0 * a->ld_out, a->ld_out,
                           a->stats_out ? a->stats_out + (m0 / 64) * k.Co * 2 : nullptr);
        if (hipGetLastError() != hipSuccess) return EVC_ELAUNCH;
    }
    return EVC_OK;
}
