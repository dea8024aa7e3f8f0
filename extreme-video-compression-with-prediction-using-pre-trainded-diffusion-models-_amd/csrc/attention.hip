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

#ifndef EVC_ATTN_MIN_WG
#define EVC_ATTN_MIN_WG 256      // fewest workgroups a launch should offer before the waves per workgroup are halved
#endif

namespace {

template <int D, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void attention_kernel(const float* __restrict__ q,
                                                                  const float* __restrict__ k,
                                                                  const float* __restrict__ v, int ld,
                                                                  float* __restrict__ out, int ld_out, int N,
                                                                  float scale) {
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
    const int q0 = blockIdx.x * 32 * WAVES + wave * 32;
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

    for (int k0 = 0; k0 < N; k0 += 32) {
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

template <int D>
int launch(const float* q, const float* k, const float* v, int ld, float* out, int ld_out, int B, int heads, int N,
           float scale, hipStream_t st) {
    const size_t lds = (size_t)2 * 32 * (D + 4) * sizeof(float);
    const long long bh = (long long)B * heads;
    int waves = 4;
    while (waves > 1 && bh * ((N + 32 * waves - 1) / (32 * waves)) < EVC_ATTN_MIN_WG) waves >>= 1;
    dim3 grid((N + 32 * waves - 1) / (32 * waves), heads, B);
    if (waves == 4)
        hipLaunchKernelGGL((attention_kernel<D, 4>), grid, dim3(256), lds, st, q, k, v, ld, out, ld_out, N, scale);
    else if (waves == 2)
        hipLaunchKernelGGL((attention_kernel<D, 2>), grid, dim3(128), lds, st, q, k, v, ld, out, ld_out, N, scale);
    else
        hipLaunchKernelGGL((attention_kernel<D, 1>), grid, dim3(64), lds, st, q, k, v, ld, out, ld_out, N, scale);
    return hipGetLastError() == hipSuccess ? EVC_OK : EVC_ELAUNCH;
}

}  // namespace

extern "C" int evc_attention_f32(const float* q, const float* k, const float* v, int ld_qkv, float* out, int ld_out,
                                 int B, int heads, int N, int D, float scale, void* stream) {
    if (!q || !k || !v || !out || B <= 0 || heads <= 0 || N <= 0 || ld_qkv < heads * D || ld_out < heads * D)
        return EVC_EINVAL;
    if ((ld_qkv & 3) || (ld_out & 3)) return EVC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    switch (D) {
        case 192: return launch<192>(q, k, v, ld_qkv, out, ld_out, B, heads, N, scale, st);
        case 64: return launch<64>(q, k, v, ld_qkv, out, ld_out, B, heads, N, scale, st);
        case 32: return launch<32>(q, k, v, ld_qkv, out, ld_out, B, heads, N, scale, st);
        default: return EVC_EUNSUPPORTED;
    }
}
