// attention.hip -- multi-head spatial self-attention, flash style, on the gfx950 f32 matrix cores.
//
//   out[b][n][h*D + d] = sum_m softmax_m(q[b][n][h] . k[b][m][h] * scale) * v[b][m][h][d]
//
// Replaces the two einsums and the softmax of AttnBlockpp.forward (reference
// models/better/layerspp.py:241-246), which materialise a (B*heads, HW, HW) score tensor.
//
// Both products are issued in "swapped" form so that the QUERY index is the MFMA lane (column):
//   S^T[key][query] = K[key][:] . Q[query][:]      A = K tile (LDS, ds_read_b128), B = Q (registers)
//   O^T[d][query]   = sum_key V[key][d] P^T[key][query]   A = V tile (LDS, ds_read_b32), B = P (the S^T
//                                                         accumulator registers, no data movement)
// so the online-softmax state (running max / sum, rescale factor) is lane-local: a lane owns one query
// and, with its partner lane ^ 32, all 32 keys of the tile.  v_mfma_f32_32x32x2_f32 is exact f32.
// One wave = 32 queries; a workgroup of WAVES waves shares each 32-key K/V tile through LDS.
#include <hip/hip_runtime.h>
#include <math.h>
#include "../../include/evc_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef EVC_ATTN_WG_TARGET
#define EVC_ATTN_WG_TARGET 512      // workgroups a key-split launch should offer (2 per CU)
#endif
#ifndef EVC_ATTN_MIN_WG
#define EVC_ATTN_MIN_WG 256      // fewest workgroups a launch should offer before the waves per workgroup are halved
#endif

namespace {

template <int D, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void attention_kernel(const float* __restrict__ q,
                                                                  const float* __restrict__ k,
                                                                  const float* __restrict__ v, int ld,
                                                                  float* __restrict__ out, int ld_out, int N,
                                                                  float scale, int kparts, int tiles_per_part,
                                                                  float* __restrict__ part_o,
                                                                  float* __restrict__ part_ml) {
    constexpr int LDK = D + 4;         // padded LDS row: (D+4)*4 bytes is an odd number of 16-B slots
    constexpr int NG = D / 8;          // 8-deep k groups of the QK^T product
    constexpr int DT = D / 32;         // 32-wide d tiles of the PV product
    constexpr int NT = 64 * WAVES;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const Ks = smem;                 // [32][LDK]
    float* const Vs = smem + 32 * LDK;      // [32][LDK]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    // key split (flash-decoding style): block x = query block * kparts + key part; each part walks its own range of
    // 32-key tiles and leaves an UNNORMALISED partial (o, running max, running sum) for attention_merge_kernel
    const int kpart = kparts > 1 ? blockIdx.x % kparts : 0;
    const int qblk = kparts > 1 ? blockIdx.x / kparts : blockIdx.x;
    const int k_begin = kpart * tiles_per_part * 32;
    const int k_end = kparts > 1 ? min(N, k_begin + tiles_per_part * 32) : N;
    const int q0 = qblk * 32 * WAVES + wave * 32;
    const size_t base = (size_t)b * N * ld + (size_t)h * D;
    const int myq = q0 + l31;
    const bool qvalid = myq < N;

    // Q fragment (B operand): lane holds Q[myq][8g + 4*half + e], e = 0..3
    float4 qf[NG];
    {
        const float* qp = q + base + (size_t)(qvalid ? myq : 0) * ld + 4 * half;
#pragma unroll
        for (int g = 0; g < NG; ++g)
            qf[g] = qvalid ? *reinterpret_cast<const float4*>(qp + 8 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
    }

    f32x16 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    for (int k0 = k_begin; k0 < k_end; k0 += 32) {
        __syncthreads();   // previous tile fully consumed
        // cooperative K/V tile load: 32 rows x D floats each
        for (int i = tid; i < 32 * (D / 4); i += NT) {
            const int row = i / (D / 4), c4 = i - row * (D / 4);
            const int key = k0 + row;
            float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
            if (key < N) {
                kv = *reinterpret_cast<const float4*>(k + base + (size_t)key * ld + 4 * c4);
                vv = *reinterpret_cast<const float4*>(v + base + (size_t)key * ld + 4 * c4);
            }
            *reinterpret_cast<float4*>(Ks + row * LDK + 4 * c4) = kv;
            *reinterpret_cast<float4*>(Vs + row * LDK + 4 * c4) = vv;
        }
        __syncthreads();

        // S^T tile: rows = keys (registers), col = query (lane)
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const float* kp = Ks + l31 * LDK + 4 * half;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const float4 a = *reinterpret_cast<const float4*>(kp + 8 * g);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qf[g].x, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qf[g].y, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, qf[g].z, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, qf[g].w, s, 0, 0, 0);
        }

        // online softmax over this tile's 32 keys (16 here, 16 in lane ^ 32)
        float m_tile = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            s[r] = key < N ? s[r] * scale : -INFINITY;
            m_tile = fmaxf(m_tile, s[r]);
        }
        m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 32));
        const float m_new = fmaxf(m_run, m_tile);      // finite: every tile holds at least one valid key
        const float alpha = __expf(m_run - m_new);     // exp(-inf) = 0 on the first tile
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = __expf(s[r] - m_new);
            psum += s[r];
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[t][r] *= alpha;

        // O^T += V^T P^T : step r consumes keys kr (lanes 0-31) and kr + 4 (lanes 32-63)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kr = (r & 3) + 8 * (r >> 2) + 4 * half;
            const float* vp = Vs + kr * LDK + l31;
#pragma unroll
            for (int t = 0; t < DT; ++t)
                o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[32 * t], s[r], o[t], 0, 0, 0);
        }
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32);
    if (kparts > 1) {
        // partial result: o unnormalised [part][b][n][h*D + d], (m, l) per [part][b][h][n]
        if (qvalid) {
            const size_t heads = gridDim.y, B = gridDim.z;
            float* op = part_o + (((size_t)kpart * B + b) * N + myq) * (heads * D) + (size_t)h * D;
#pragma unroll
            for (int t = 0; t < DT; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(op + 32 * t + 8 * g + 4 * half) =
                        make_float4(o[t][4 * g], o[t][4 * g + 1], o[t][4 * g + 2], o[t][4 * g + 3]);
            if (half == 0) {
                float* ml = part_ml + ((((size_t)kpart * B + b) * heads + h) * N + myq) * 2;
                ml[0] = m_run; ml[1] = l_tot;
            }
        }
        return;
    }
    const float inv = 1.0f / l_tot;
    if (qvalid) {
        float* op = out + ((size_t)b * N + myq) * ld_out + (size_t)h * D;
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 w = make_float4(o[t][4 * g] * inv, o[t][4 * g + 1] * inv, o[t][4 * g + 2] * inv,
                                       o[t][4 * g + 3] * inv);
                *reinterpret_cast<float4*>(op + 32 * t + 8 * g + 4 * half) = w;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same attention on the fp16 matrix cores ("f16x3", see csrc/conv_igemm.hip): every operand -- Q, K, V and the
// softmax weights P -- is scaled by a power of two into fp16's range (Q / K / V from an element bound the caller derives
// from the q|k|v tensor's moments, P in [0, 1] by 2^10), split 2-way into fp16 (2^-22 relative) and each product runs as
// three v_mfma_f32_32x32x16_f16 with fp32 accumulation: 72 MFMAs of 32 cycles per 32-key tile and wave instead of 192 of
// 64 cycles on the f32 pipe.  Same orientation as attention_kernel (S^T = K Q^T, O^T = V^T P^T: the query is the lane,
// softmax state lane-local).  P feeds the second product straight from the S^T accumulator registers: registers
// 8s..8s+7 of a 32x32 accumulator are the B fragment of k-step s with the key order permuted to
// key = 16s + 8(j>>2) + 4*half + (j&3); V is staged TRANSPOSED ([d][key position], fp16 planes) with the keys stored in
// exactly that order, so its A fragment is one 16-byte LDS read.  K is staged [key][d].  Row pitches are an odd number of
// 16-byte slots (conflict-free ds_read_b128).
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float pow2_scale_for(float bound_sq_bits_as_float, int target_exp) {
    const float b = sqrtf(bound_sq_bits_as_float);
    if (!(b < 3.0e38f)) return __builtin_nanf("");      // non-finite tensor (NaN pattern from the bound kernels): NaN out
    return b > 0.f ? ldexpf(1.0f, target_exp - ilogbf(b)) : 1.0f;
}

// the operands are scaled from true element bounds (<= 2^10), so nothing can overflow and nothing is clamped
__device__ __forceinline__ void split2h(float v, _Float16& a, _Float16& b) {
    a = (_Float16)v;
    b = (_Float16)(v - (float)a);
}

typedef __attribute__((address_space(3))) void attn_lds_void;
typedef const __attribute__((address_space(1))) void attn_glb_void;

// PRE = true: K / V arrive already scaled, split and laid out -- one contiguous image per (sample, head, 32-key tile) in exactly
// the LDS layout below, written once per launch by attention_kv_planes_kernel -- and a tile is staged by LDS-DMA
// (global_load_lds_dwordx4: no registers, no conversion arithmetic, no ds_write) into one of TWO LDS images, the copy of tile
// t + 1 running under the MFMAs of tile t with a single barrier per tile.  Same numbers as PRE = false (the conversion is the
// same code, run once per tile instead of once per tile and query block).
template <int D, int WAVES, bool PRE = false>
__global__ __launch_bounds__(64 * WAVES, 1) void attention_f16_kernel(const float* __restrict__ q,
                                                                      const float* __restrict__ k,
                                                                      const float* __restrict__ v, int ld,
                                                                      float* __restrict__ out, int ld_out, int N,
                                                                      float scale, const unsigned* __restrict__ bounds,
                                                                      int kparts, int tiles_per_part,
                                                                      float* __restrict__ part_o,
                                                                      float* __restrict__ part_ml,
                                                                      const char* __restrict__ kv_planes) {
    constexpr int NKS = D / 16;            // k-steps of the QK^T product
    constexpr int DT = D / 32;             // 32-wide d tiles of the PV product
    constexpr int NT = 64 * WAVES;
    constexpr int KROW = 2 * D + 16;       // bytes per key row of a K plane: D/8 + 1 slots (odd)
    constexpr int VROW = 64 + 16;          // bytes per d row of a V^T plane: 32 key positions + 1 slot
    constexpr int KPL = 32 * KROW, VPL = D * VROW;
    constexpr int TILE_BYTES = 2 * KPL + 2 * VPL;
    static_assert(!PRE || WAVES == 4, "the pre-converted path stages with 256 threads");
    extern __shared__ __attribute__((aligned(16))) char smem_h[];
    char* Ks = smem_h;                     // [2 planes][32 keys][KROW]      (PRE: + a second image behind the first)
    char* Vs = smem_h + 2 * KPL;           // [2 planes][D][VROW]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int kpart = kparts > 1 ? blockIdx.x % kparts : 0;
    const int qblk = kparts > 1 ? blockIdx.x / kparts : blockIdx.x;
    const int k_begin = kpart * tiles_per_part * 32;
    const int k_end = kparts > 1 ? min(N, k_begin + tiles_per_part * 32) : N;
    const int q0 = qblk * 32 * WAVES + wave * 32;
    const size_t base = (size_t)b * N * ld + (size_t)h * D;
    const int myq = q0 + l31;
    const bool qvalid = myq < N;

    // operand scales (powers of two): bound * scale in [2^9, 2^10) -> at most 1024, far above fp16's subnormals
    const float sq = pow2_scale_for(__uint_as_float(bounds[0]), 9);
    const float sk = pow2_scale_for(__uint_as_float(bounds[1]), 9);
    const float sv = pow2_scale_for(__uint_as_float(bounds[2]), 9);
    constexpr float SP = 1024.0f;                     // softmax weights are in [0, 1]
    const float s_mul = scale / (sq * sk);            // accumulator -> logits
    const float o_mul = 1.0f / (SP * sv);             // accumulator -> sum_k p_k v_k

    // Q fragments (B operand of S^T = K Q^T): lane (query l31, half) holds Q[q][16s + 8*half + j], both fp16 planes
    h16x8 q1[NKS], q2[NKS];
    {
        const float* qp = q + base + (size_t)(qvalid ? myq : 0) * ld + 8 * half;
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
            if (qvalid) {
                a = *reinterpret_cast<const float4*>(qp + 16 * s);
                c = *reinterpret_cast<const float4*>(qp + 16 * s + 4);
            }
            const float x[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) { _Float16 u, w; split2h(x[j] * sq, u, w); q1[s][j] = u; q2[s][j] = w; }
        }
    }

    f32x16 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // LDS fragment offsets (bytes)
    const int k_rd = l31 * KROW + 16 * half;                      // + plane * KPL + 32 * s
    const int v_rd = l31 * VROW + 16 * half;                      // + plane * VPL + 32 * t * VROW + 32 * s2

    // K / V staging with a register prefetch: the global loads of tile t+1 are issued before the MFMAs of tile t and
    // converted (scale, clamp, 2-way fp16 split) + written to LDS after them, so their latency hides behind the tile's
    // compute.  K unit = (key, 8-channel chunk): two float4 loads -> two 16-byte LDS writes (one per plane);
    // V unit = (group of 4 consecutive keys, channel d), lanes along d (coalesced rows) -> two 8-byte writes into the
    // transposed image at the keys' permuted positions.
    constexpr int KU = (32 * (D / 8) + NT - 1) / NT;          // K units per thread
    constexpr int VU = (8 * D + NT - 1) / NT;                 // V units per thread
    float4 kreg[KU][2];
    float vreg[VU][4];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < KU; ++i) {
            const int u = tid + NT * i;
            const int row = u / (D / 8), ch = u - row * (D / 8);
            const int key = k0 + row;
            kreg[i][0] = kreg[i][1] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (u < 32 * (D / 8) && key < N) {
                const float* kp = k + base + (size_t)key * ld + 8 * ch;
                kreg[i][0] = *reinterpret_cast<const float4*>(kp);
                kreg[i][1] = *reinterpret_cast<const float4*>(kp + 4);
            }
        }
#pragma unroll
        for (int i = 0; i < VU; ++i) {
            const int u = tid + NT * i;
            const int g = u / D, d = u - g * D;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int key = k0 + 4 * g + e;
                vreg[i][e] = (u < 8 * D && key < N) ? v[base + (size_t)key * ld + d] : 0.f;
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < KU; ++i) {
            const int u = tid + NT * i;
            if (u >= 32 * (D / 8)) break;
            const int row = u / (D / 8), ch = u - row * (D / 8);
            const float x[8] = {kreg[i][0].x, kreg[i][0].y, kreg[i][0].z, kreg[i][0].w,
                                kreg[i][1].x, kreg[i][1].y, kreg[i][1].z, kreg[i][1].w};
            h16x8 p1, p2;
#pragma unroll
            for (int j = 0; j < 8; ++j) { _Float16 uu, ww; split2h(x[j] * sk, uu, ww); p1[j] = uu; p2[j] = ww; }
            char* dst = Ks + row * KROW + 16 * ch;
            *reinterpret_cast<h16x8*>(dst) = p1;
            *reinterpret_cast<h16x8*>(dst + KPL) = p2;
        }
#pragma unroll
        for (int i = 0; i < VU; ++i) {
            const int u = tid + NT * i;
            if (u >= 8 * D) break;
            const int g = u / D, d = u - g * D;
            h16x4 p1, p2;
#pragma unroll
            for (int e = 0; e < 4; ++e) { _Float16 uu, ww; split2h(vreg[i][e] * sv, uu, ww); p1[e] = uu; p2[e] = ww; }
            // keys 4g .. 4g+3 sit at positions 16*(g>>2) + 8*(g&1) + 4*((g>>1)&1) + (0..3)
            const int pos = 16 * (g >> 2) + 8 * (g & 1) + 4 * ((g >> 1) & 1);
            char* dst = Vs + d * VROW + 2 * pos;
            *reinterpret_cast<h16x4*>(dst) = p1;
            *reinterpret_cast<h16x4*>(dst + VPL) = p2;
        }
    };

    // PRE: LDS-DMA of one tile image, 16 bytes per lane, 1 KiB per wave and instruction (wave w moves bytes
    // [4096 j + 1024 w, + 1024) of the image for j = 0, 1, ...)
    const char* const tile0 = PRE ? kv_planes + ((size_t)(b * gridDim.y + h) * ((N + 31) / 32)) * TILE_BYTES : nullptr;
    auto dma_tile = [&](int k0, int buf) {
        const char* src = tile0 + (size_t)(k0 >> 5) * TILE_BYTES + wave * 1024 + lane * 16;
        char* dst = smem_h + buf * TILE_BYTES + wave * 1024;
#pragma unroll
        for (int j = 0; j < (TILE_BYTES + 4095) / 4096; ++j)
            if (4096 * j + 1024 * wave < TILE_BYTES)            // wave-uniform (TILE_BYTES is a multiple of 1024)
                __builtin_amdgcn_global_load_lds((attn_glb_void*)(src + 4096 * j), (attn_lds_void*)(dst + 4096 * j), 16, 0, 0);
    };
    static_assert(TILE_BYTES % 1024 == 0, "tile image = whole wave-instructions");
    if (k_begin < k_end) {
        if constexpr (PRE) dma_tile(k_begin, 0); else load_tile(k_begin);
    }
    for (int k0 = k_begin; k0 < k_end; k0 += 32) {
        if constexpr (PRE) {
            const int buf = ((k0 - k_begin) >> 5) & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the tile has landed ...
            __syncthreads();                                    // ... and everybody's; the other image is fully consumed
            if (k0 + 32 < k_end) dma_tile(k0 + 32, buf ^ 1);    // copied during this tile's MFMAs
            Ks = smem_h + buf * TILE_BYTES;
            Vs = Ks + 2 * KPL;
        } else {
            __syncthreads();   // previous tile fully consumed
            store_tile();
            __syncthreads();
            if (k0 + 32 < k_end) load_tile(k0 + 32);       // in flight during this tile's MFMAs
        }

        // ---- S^T tile: rows = keys (registers), col = query (lane) ----
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int st = 0; st < NKS; ++st) {
            const h16x8 k1 = *reinterpret_cast<const h16x8*>(Ks + k_rd + 32 * st);
            const h16x8 k2 = *reinterpret_cast<const h16x8*>(Ks + KPL + k_rd + 32 * st);
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(k2, q1[st], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1, q2[st], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1, q1[st], s, 0, 0, 0);
        }

        // ---- online softmax over this tile's 32 keys (16 here, 16 in lane ^ 32) ----
        // LAZY reference: the running reference m_run only moves when a tile's maximum exceeds it by more than LAZY_TAU, so the
        // rescale of the 6 x 16 output accumulators (a wave-uniform branch any of the 32 queries can trigger) runs on a few
        // tiles instead of on nearly all of them; in between the weights are exp(s - m_run) <= e^LAZY_TAU = 33 instead of <= 1,
        // which the 2^10 scale of P still keeps inside fp16's range (33 * 1024 < 65504) at the same 2^-22 split accuracy.
        // The normalisation (l_run) and the key-part merge use the same reference, so the result is the same softmax.
        constexpr float LAZY_TAU = 3.5f;
        float m_tile = -INFINITY;
        if (k0 + 32 > N) {                             // wave-uniform: only a ragged last tile has keys to mask
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                s[r] = key < N ? s[r] * s_mul : -INFINITY;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] *= s_mul;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) m_tile = fmaxf(m_tile, s[r]);
        m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 32));           // finite: every tile holds at least one valid key
        const float m_new = m_tile > m_run + LAZY_TAU ? m_tile : m_run;     // first tile: m_run = -inf -> m_tile
        const float alpha = __expf(m_run - m_new);     // exp(-inf) = 0 on the first tile, exactly 1 while the reference stays
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = __expf(s[r] - m_new);
            psum += s[r];
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
        if (__any(alpha != 1.0f)) {                    // wave-uniform: only when some query's reference moved
#pragma unroll
            for (int t = 0; t < DT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
        }

        // ---- O^T += V^T P^T: k-step s2 takes accumulator registers 8*s2 .. 8*s2+7 as its B fragment ----
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            h16x8 p1, p2;
#pragma unroll
            for (int j = 0; j < 8; ++j) { _Float16 uu, ww; split2h(s[8 * s2 + j] * SP, uu, ww); p1[j] = uu; p2[j] = ww; }
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                const h16x8 v1 = *reinterpret_cast<const h16x8*>(Vs + v_rd + 32 * t * VROW + 32 * s2);
                const h16x8 v2 = *reinterpret_cast<const h16x8*>(Vs + VPL + v_rd + 32 * t * VROW + 32 * s2);
                o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v2, p1, o[t], 0, 0, 0);
                o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1, p2, o[t], 0, 0, 0);
                o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1, p1, o[t], 0, 0, 0);
            }
        }
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32);
    if (kparts > 1) {
        if (qvalid) {
            const size_t heads = gridDim.y, B = gridDim.z;
            float* op = part_o + (((size_t)kpart * B + b) * N + myq) * (heads * D) + (size_t)h * D;
#pragma unroll
            for (int t = 0; t < DT; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(op + 32 * t + 8 * g + 4 * half) =
                        make_float4(o[t][4 * g] * o_mul, o[t][4 * g + 1] * o_mul, o[t][4 * g + 2] * o_mul, o[t][4 * g + 3] * o_mul);
            if (half == 0) {
                float* ml = part_ml + ((((size_t)kpart * B + b) * heads + h) * N + myq) * 2;
                ml[0] = m_run; ml[1] = l_tot;
            }
        }
        return;
    }
    const float inv = o_mul / l_tot;
    if (qvalid) {
        float* op = out + ((size_t)b * N + myq) * ld_out + (size_t)h * D;
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 w = make_float4(o[t][4 * g] * inv, o[t][4 * g + 1] * inv, o[t][4 * g + 2] * inv,
                                       o[t][4 * g + 3] * inv);
                *reinterpret_cast<float4*>(op + 32 * t + 8 * g + 4 * half) = w;
            }
    }
}

// K / V of one launch -> the tile images attention_f16_kernel<D, 4, true> stages by LDS-DMA: per (sample, head, 32-key tile)
// [2 planes][32 keys][KROW] of K and [2 planes][D][VROW] of V^T (keys in the permuted order of the PV product's B fragment),
// scaled by the same powers of two and split into the same two fp16 planes as the in-kernel conversion (split2h).  The row
// pads are never read.  grid (tiles, heads, B), block 256.
template <int D>
__global__ __launch_bounds__(256) void attention_kv_planes_kernel(const float* __restrict__ k, const float* __restrict__ v,
                                                                  int ld, int N, const unsigned* __restrict__ bounds,
                                                                  char* __restrict__ planes) {
    constexpr int NT = 256;
    constexpr int KROW = 2 * D + 16, VROW = 64 + 16;
    constexpr int KPL = 32 * KROW, VPL = D * VROW;
    constexpr int TILE_BYTES = 2 * KPL + 2 * VPL;
    // the image is assembled in LDS (the transposed V^T pieces are 8-byte writes 80 bytes apart) and leaves as one contiguous copy
    extern __shared__ __attribute__((aligned(16))) char img[];
    const int tid = threadIdx.x, tile = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int k0 = tile * 32;
    const size_t base = (size_t)b * N * ld + (size_t)h * D;
    char* const Ks = img;
    char* const Vs = Ks + 2 * KPL;
    for (int i = tid; i < 2 * 32 + 2 * D; i += NT) {              // the row pads are copied too: keep them defined
        char* row = i < 64 ? Ks + i * KROW + 2 * D : Vs + (i - 64) * VROW + 64;
        *reinterpret_cast<float4*>(row) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float sk = pow2_scale_for(__uint_as_float(bounds[1]), 9);
    const float sv = pow2_scale_for(__uint_as_float(bounds[2]), 9);
    // all loads first (they are independent: one memory round trip instead of one per unit), then convert + place
    constexpr int KU = (32 * (D / 8) + NT - 1) / NT;             // K unit = (key, 8-channel chunk)
    constexpr int VU = (8 * D + NT - 1) / NT;                     // V unit = (4 consecutive keys, channel d), lanes along d
    float4 kreg[KU][2];
    float vreg[VU][4];
#pragma unroll
    for (int i = 0; i < KU; ++i) {
        const int u = tid + NT * i;
        const int row = u / (D / 8), ch = u - row * (D / 8);
        const int key = k0 + row;
        kreg[i][0] = kreg[i][1] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (u < 32 * (D / 8) && key < N) {
            const float* kp = k + base + (size_t)key * ld + 8 * ch;
            kreg[i][0] = *reinterpret_cast<const float4*>(kp);
            kreg[i][1] = *reinterpret_cast<const float4*>(kp + 4);
        }
    }
#pragma unroll
    for (int i = 0; i < VU; ++i) {
        const int u = tid + NT * i;
        const int g = u / D, d = u - g * D;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int key = k0 + 4 * g + e;
            vreg[i][e] = (u < 8 * D && key < N) ? v[base + (size_t)key * ld + d] : 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < KU; ++i) {
        const int u = tid + NT * i;
        if (u >= 32 * (D / 8)) break;
        const int row = u / (D / 8), ch = u - row * (D / 8);
        const float x[8] = {kreg[i][0].x, kreg[i][0].y, kreg[i][0].z, kreg[i][0].w, kreg[i][1].x, kreg[i][1].y, kreg[i][1].z, kreg[i][1].w};
        h16x8 p1, p2;
#pragma unroll
        for (int j = 0; j < 8; ++j) { _Float16 uu, ww; split2h(x[j] * sk, uu, ww); p1[j] = uu; p2[j] = ww; }
        char* dst = Ks + row * KROW + 16 * ch;
        *reinterpret_cast<h16x8*>(dst) = p1;
        *reinterpret_cast<h16x8*>(dst + KPL) = p2;
    }
#pragma unroll
    for (int i = 0; i < VU; ++i) {
        const int u = tid + NT * i;
        if (u >= 8 * D) break;
        const int g = u / D, d = u - g * D;
        h16x4 p1, p2;
#pragma unroll
        for (int e = 0; e < 4; ++e) { _Float16 uu, ww; split2h(vreg[i][e] * sv, uu, ww); p1[e] = uu; p2[e] = ww; }
        const int pos = 16 * (g >> 2) + 8 * (g & 1) + 4 * ((g >> 1) & 1);
        char* dst = Vs + d * VROW + 2 * pos;
        *reinterpret_cast<h16x4*>(dst) = p1;
        *reinterpret_cast<h16x4*>(dst + VPL) = p2;
    }
    __syncthreads();
    float4* out = reinterpret_cast<float4*>(planes + ((size_t)(b * gridDim.y + h) * gridDim.x + tile) * TILE_BYTES);
    for (int i = tid; i < TILE_BYTES / 16; i += NT) out[i] = reinterpret_cast<const float4*>(img)[i];
}

template <int D>
constexpr long long attention_tile_bytes() { return 2LL * 32 * (2 * D + 16) + 2LL * D * 80; }

// out[b][n][h*D + d] = sum_p w_p * o_p[d] / sum_p w_p * l_p,  w_p = exp(m_p - max_p m_p): the exact online-softmax
// merge of the key parts (same formula the kernel applies tile by tile).  One thread per (b, n, h, 4 channels).
__global__ void attention_merge_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml,
                                       float* __restrict__ out, int ld_out, int B, int heads, int N, int D,
                                       int kparts) {
    const int D4 = D >> 2;
    const size_t total = (size_t)B * N * heads * D4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int d4 = (int)(i % D4);
        size_t t = i / D4;
        const int h = (int)(t % heads); t /= heads;
        const int n = (int)(t % N);
        const int b = (int)(t / N);
        float m = -INFINITY;
        for (int p = 0; p < kparts; ++p) m = fmaxf(m, part_ml[((((size_t)p * B + b) * heads + h) * N + n) * 2]);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        float l = 0.f;
        for (int p = 0; p < kparts; ++p) {
            const float* ml = part_ml + ((((size_t)p * B + b) * heads + h) * N + n) * 2;
            const float w = __expf(ml[0] - m);
            l += w * ml[1];
            const float4 v = *reinterpret_cast<const float4*>(part_o + (((size_t)p * B + b) * N + n) * ((size_t)heads * D) +
                                                              (size_t)h * D + 4 * d4);
            acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
        }
        const float inv = 1.0f / l;
        *reinterpret_cast<float4*>(out + ((size_t)b * N + n) * ld_out + (size_t)h * D + 4 * d4) =
            make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
    }
}

struct AttnPlan { int waves, kparts, tiles_per_part; };

// Waves per workgroup as before; then, when the launch would leave SIMDs idle (fewer than ~1024 waves) and the key
// loop is long enough, split the keys so that every wave's serial chain shortens (each part >= 2 tiles).
AttnPlan attention_plan(int B, int heads, int N, bool have_ws) {
    AttnPlan p;
    const long long bh = (long long)B * heads;
    const int ntiles = (N + 31) / 32;
    if (!have_ws) {
        p.waves = 4;
        while (p.waves > 1 && bh * ((N + 32 * p.waves - 1) / (32 * p.waves)) < EVC_ATTN_MIN_WG) p.waves >>= 1;
        p.kparts = 1;
    } else {
        // (waves per workgroup, key parts) jointly: the largest workgroup (K/V tile shared by more waves, fewer LDS
        // footprints) and the fewest key parts that still offer EVC_ATTN_WG_TARGET workgroups; failing that, the
        // combination with the most workgroups.  Every key part keeps at least 2 tiles.
        long long best = -1;
        p.waves = 1; p.kparts = 1;
        bool done = false;
        for (int w = 4; w >= 1 && !done; w >>= 1)
            for (int kp = 1; kp <= 4; kp *= 2) {
                if (kp > 1 && ntiles / kp < 2) break;
                const long long wgs = bh * ((N + 32 * w - 1) / (32 * w)) * kp;
                if (wgs >= EVC_ATTN_WG_TARGET) { p.waves = w; p.kparts = kp; done = true; break; }
                if (wgs > best) { best = wgs; p.waves = w; p.kparts = kp; }
            }
    }
    p.tiles_per_part = (ntiles + p.kparts - 1) / p.kparts;
    p.kparts = (ntiles + p.tiles_per_part - 1) / p.tiles_per_part;       // no empty parts
    return p;
}

// The fp16 kernel converts every K / V tile on the fly (scale, split, transpose), a cost shared by the waves of a
// workgroup: always the largest workgroup the query count fills, then key parts until the launch offers ~2 per CU.
AttnPlan attention_plan_f16(int B, int heads, int N) {
    AttnPlan p;
    const long long bh = (long long)B * heads;
    const int ntiles = (N + 31) / 32;
    p.waves = N >= 128 ? 4 : (N >= 64 ? 2 : 1);
    const long long qblocks = (N + 32 * p.waves - 1) / (32 * p.waves);
    // key parts: the count that minimises rounds x tiles per part (two 56-KB workgroups fit a CU: 512 slots), each part
    // charged 0.3 tile-times for its share of the merge.  B=9, 32x32, 2 heads: 144 query blocks -> 3 parts (432 workgroups,
    // one round of 11 tiles) instead of 4 (576 workgroups = 2 rounds of 8).
    p.kparts = 1;
    double best = 1e30;
    for (int kp = 1; kp <= 8; ++kp) {
        if (kp > 1 && ntiles / kp < 2) break;
        const long long wgs = bh * qblocks * kp;
        const double cost = (double)((wgs + EVC_ATTN_WG_TARGET - 1) / EVC_ATTN_WG_TARGET) * ((ntiles + kp - 1) / kp) + (kp > 1 ? 0.3 * kp : 0.0);
        if (cost < best - 1e-9) { best = cost; p.kparts = kp; }
    }
    p.tiles_per_part = (ntiles + p.kparts - 1) / p.kparts;
    p.kparts = (ntiles + p.tiles_per_part - 1) / p.tiles_per_part;
    return p;
}

int g_attn_pre = 1;       // run-time switch (evc_attention_set_option "kv_planes"): K / V converted once per launch (A/B)
int g_attn_f16_min_keys = 128;   // run-time switch "f16_min_keys": fewest keys for which the fp16-split kernel is used

// bytes of the key-part buffers at the head of the workspace (part_o | part_ml), rounded to 16
long long attention_parts_bytes(int B, int heads, int N, int D) {
    const int kp = attention_plan(B, heads, N, true).kparts, kp16 = attention_plan_f16(B, heads, N).kparts;
    const int kparts = kp > kp16 ? kp : kp16;
    if (kparts <= 1) return 0;
    const long long n = (long long)kparts * B * N * heads * ((long long)D + 2) * (long long)sizeof(float);
    return (n + 15) / 16 * 16;
}

// attention_kernel<256, *> stages 2 x 32 x 260 floats = 65 KB: above the 64 KB a kernel gets without asking
template <int D, int WAVES>
int launch_f32(dim3 grid, size_t lds, hipStream_t st, const float* q, const float* k, const float* v, int ld, float* out, int ld_out,
               int N, float scale, int kparts, int tiles_per_part, float* part_o, float* part_ml) {
    if (lds > 64 * 1024) {
        static unsigned long long done = 0;                       // one bit per device id, set after the call succeeded
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return EVC_ELAUNCH;
        const unsigned long long bit = 1ULL << (dev & 63);
        if (!(__atomic_load_n(&done, __ATOMIC_ACQUIRE) & bit)) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_kernel<D, WAVES>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return EVC_ELAUNCH;
            __atomic_fetch_or(&done, bit, __ATOMIC_RELEASE);
        }
    }
    hipLaunchKernelGGL((attention_kernel<D, WAVES>), grid, dim3(64 * WAVES), lds, st, q, k, v, ld, out, ld_out, N, scale, kparts,
                       tiles_per_part, part_o, part_ml);
    return EVC_OK;
}

template <int D>
int launch(const float* q, const float* k, const float* v, int ld, float* out, int ld_out, int B, int heads, int N,
           float scale, float* ws, const unsigned* bounds, hipStream_t st) {
    const size_t lds = (size_t)2 * 32 * (D + 4) * sizeof(float);
    // head widths above 192 stay on the f32 kernel: the fp16 kernel keeps Q (both planes) and O in registers, D/2 + D/2 of them
    constexpr bool F16_OK = D <= 192;
    const bool use_f16 = F16_OK && bounds && N >= g_attn_f16_min_keys;
    const AttnPlan pl = (use_f16 && ws) ? attention_plan_f16(B, heads, N) : attention_plan(B, heads, N, ws != nullptr);
    const int waves = pl.waves;
    float* part_o = ws;
    float* part_ml = ws ? ws + (size_t)pl.kparts * B * N * heads * D : nullptr;
    dim3 grid(((N + 32 * waves - 1) / (32 * waves)) * pl.kparts, heads, B);
    // tiny key sets (N < 128: the 8x8 level) stay on the f32 kernel: with two key tiles there is nothing to amortise the
    // fp16 conversion of K / V over (measured B=9, 8x8: 26 us f32 vs 39 us fp16 split; 16x16: 81 vs 29; 32x32: 260 vs 155)
    if constexpr (F16_OK) if (use_f16) {
        const size_t lds16 = (size_t)attention_tile_bytes<D>();
        // with a workspace: K / V converted once per launch into tile images (behind the key-part buffers), staged by LDS-DMA
        // (from 512 keys on: below that the pre-pass launch costs more than the conversions it saves -- B = 9, 256 keys, 3 heads:
        // 36 us with it, 29 without; 1024 keys, 2 heads: 110 vs 127 -- profiles/r04_attn_kv_planes_ab.log)
        if (waves == 4 && ws && g_attn_pre && N >= 512) {
            const int ntiles = (N + 31) / 32;
            char* planes = reinterpret_cast<char*>(ws) + attention_parts_bytes(B, heads, N, D);
            hipLaunchKernelGGL((attention_kv_planes_kernel<D>), dim3(ntiles, heads, B), dim3(256), lds16, st, k, v, ld, N, bounds, planes);
            if (2 * lds16 > 64 * 1024) {
                static unsigned long long done = 0;                   // one bit per device id
                int dev = 0;
                if (hipGetDevice(&dev) != hipSuccess) return EVC_ELAUNCH;
                const unsigned long long bit = 1ULL << (dev & 63);
                if (!(__atomic_load_n(&done, __ATOMIC_ACQUIRE) & bit)) {
                    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_f16_kernel<D, 4, true>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * lds16)) != hipSuccess) return EVC_ELAUNCH;
                    __atomic_fetch_or(&done, bit, __ATOMIC_RELEASE);
                }
            }
            hipLaunchKernelGGL((attention_f16_kernel<D, 4, true>), grid, dim3(256), 2 * lds16, st, q, k, v, ld, out, ld_out, N, scale,
                               bounds, pl.kparts, pl.tiles_per_part, part_o, part_ml, planes);
        } else if (waves == 4)
            hipLaunchKernelGGL((attention_f16_kernel<D, 4>), grid, dim3(256), lds16, st, q, k, v, ld, out, ld_out, N, scale,
                               bounds, pl.kparts, pl.tiles_per_part, part_o, part_ml, nullptr);
        else if (waves == 2)
            hipLaunchKernelGGL((attention_f16_kernel<D, 2>), grid, dim3(128), lds16, st, q, k, v, ld, out, ld_out, N, scale,
                               bounds, pl.kparts, pl.tiles_per_part, part_o, part_ml, nullptr);
        else
            hipLaunchKernelGGL((attention_f16_kernel<D, 1>), grid, dim3(64), lds16, st, q, k, v, ld, out, ld_out, N, scale,
                               bounds, pl.kparts, pl.tiles_per_part, part_o, part_ml, nullptr);
    }
    if (!use_f16) {
        const int rc = waves == 4 ? launch_f32<D, 4>(grid, lds, st, q, k, v, ld, out, ld_out, N, scale, pl.kparts, pl.tiles_per_part, part_o, part_ml)
                     : waves == 2 ? launch_f32<D, 2>(grid, lds, st, q, k, v, ld, out, ld_out, N, scale, pl.kparts, pl.tiles_per_part, part_o, part_ml)
                                  : launch_f32<D, 1>(grid, lds, st, q, k, v, ld, out, ld_out, N, scale, pl.kparts, pl.tiles_per_part, part_o, part_ml);
        if (rc != EVC_OK) return rc;
    }
    if (hipGetLastError() != hipSuccess) return EVC_ELAUNCH;
    if (pl.kparts > 1) {
        const size_t total = (size_t)B * N * heads * (D / 4);
        const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
        hipLaunchKernelGGL(attention_merge_kernel, dim3(blocks), dim3(256), 0, st, part_o, part_ml, out, ld_out, B, heads, N,
                           D, pl.kparts);
        if (hipGetLastError() != hipSuccess) return EVC_ELAUNCH;
    }
    return EVC_OK;
}

}  // namespace

static int attention_dispatch(const float* q, const float* k, const float* v, int ld_qkv, float* out, int ld_out,
                              int B, int heads, int N, int D, float scale, float* ws, const unsigned* bounds,
                              void* stream) {
    if (!q || !k || !v || !out || B <= 0 || heads <= 0 || N <= 0 || ld_qkv < heads * D || ld_out < heads * D)
        return EVC_EINVAL;
    if ((ld_qkv & 3) || (ld_out & 3)) return EVC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    switch (D) {
        case 256: return launch<256>(q, k, v, ld_qkv, out, ld_out, B, heads, N, scale, ws, bounds, st);
        case 192: return launch<192>(q, k, v, ld_qkv, out, ld_out, B, heads, N, scale, ws, bounds, st);
        case 128: return launch<128>(q, k, v, ld_qkv, out, ld_out, B, heads, N, scale, ws, bounds, st);
        case 64: return launch<64>(q, k, v, ld_qkv, out, ld_out, B, heads, N, scale, ws, bounds, st);
        case 32: return launch<32>(q, k, v, ld_qkv, out, ld_out, B, heads, N, scale, ws, bounds, st);
        default: return EVC_EUNSUPPORTED;
    }
}

extern "C" int evc_attention_f32(const float* q, const float* k, const float* v, int ld_qkv, float* out, int ld_out,
                                 int B, int heads, int N, int D, float scale, void* stream) {
    return attention_dispatch(q, k, v, ld_qkv, out, ld_out, B, heads, N, D, scale, nullptr, nullptr, stream);
}

extern "C" long long evc_attention_workspace_bytes(int B, int heads, int N, int D) {
    if (B <= 0 || heads <= 0 || N <= 0 || D <= 0) return EVC_EINVAL;
    long long n = attention_parts_bytes(B, heads, N, D);          // key-part buffers: one size serves both kernels
    // + the K / V tile images of the fp16 kernel (evc_attention_f16x3_f32 with >= 128 keys and a head width it covers)
    if (N >= 128 && D <= 192 && (D == 32 || D == 64 || D == 128 || D == 192))
        n += (long long)B * heads * ((N + 31) / 32) * (2LL * 32 * (2 * D + 16) + 2LL * D * 80);
    return n;
}

extern "C" int evc_attention_set_option(const char* name, int value) {
    if (!name) return EVC_EINVAL;
    const auto is = [&](const char* s) { int i = 0; while (s[i] && s[i] == name[i]) ++i; return s[i] == 0 && name[i] == 0; };
    if (is("kv_planes")) { g_attn_pre = value; return EVC_OK; }
    if (is("f16_min_keys")) { g_attn_f16_min_keys = value; return EVC_OK; }
    return EVC_EINVAL;
}

extern "C" int evc_attention_ws_f32(const float* q, const float* k, const float* v, int ld_qkv, float* out, int ld_out,
                                    int B, int heads, int N, int D, float scale, float* ws, void* stream) {
    if (evc_attention_workspace_bytes(B, heads, N, D) > 0 && !ws) return EVC_EINVAL;
    return attention_dispatch(q, k, v, ld_qkv, out, ld_out, B, heads, N, D, scale, ws, nullptr, stream);
}

extern "C" int evc_attention_f16x3_f32(const float* q, const float* k, const float* v, int ld_qkv, float* out, int ld_out,
                                       int B, int heads, int N, int D, float scale, const unsigned* bounds, float* ws,
                                       void* stream) {
    if (!bounds) return EVC_EINVAL;
    if (evc_attention_workspace_bytes(B, heads, N, D) > 0 && !ws) return EVC_EINVAL;
    return attention_dispatch(q, k, v, ld_qkv, out, ld_out, B, heads, N, D, scale, ws, bounds, stream);
}
