// split_numerics.hip -- accuracy of fp32 emulation on the bf16 matrix cores (gfx950), measured on the device.
//
//   C[M][N] = A[M][K] * B[K][N],  M = N = 128 tiles of 32x32, one wave per tile, operands read straight from
//   global memory in MFMA fragment layout.  Variants:
//     0  v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate)            -- what conv_igemm_kernel uses
//     1  3-way bf16 split (x = x1 + x2 + x3 exactly), 6 products (x1y1, x1y2, x2y1, x1y3, x2y2, x3y1),
//        v_mfma_f32_32x32x16_bf16, fp32 accumulate                                -- "bf16x6"
//     2  same split, 3 products (x1y1, x1y2, x2y1)                                -- "bf16x3" (for contrast only)
//     3  same split, all 9 products                                               -- "bf16x9"
//     4  2-way fp16 split (x*S = h1 + h2, 11 + 11 significand bits), 3 products (h1g1, h1g2, h2g1),
//        v_mfma_f32_32x32x16_f16, fp32 accumulate; A scaled by 2^3 (GroupNorm-ed activations are O(1)), B by the
//        power of two that brings max|B| into [2^14, 2^15); the result is multiplied by the exact inverse     -- "f16x3"
//     5  same split, all 4 products                                               -- "f16x4"
//     6  variant 4 without any scaling (shows why the scaling is needed: h2 falls into fp16 subnormals)
//   Each is compared with an fp64 host reference; errors are relative to max|C|.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/split_numerics tools/split_numerics.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(const float (&x)[8], bf16x8& p1, bf16x8& p2, bf16x8& p3) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 a = (__bf16)x[j];
        const float r = x[j] - (float)a;
        const __bf16 b = (__bf16)r;
        const float r2 = r - (float)b;
        p1[j] = a; p2[j] = b; p3[j] = (__bf16)r2;
    }
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split2h(const float (&x)[8], float s, f16x8& p1, f16x8& p2) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = fminf(fmaxf(x[j] * s, -65504.f), 65504.f);
        const _Float16 a = (_Float16)v;
        p1[j] = a; p2[j] = (_Float16)(v - (float)a);
    }
}

template <int VAR>
__global__ __launch_bounds__(64) void gemm_kernel(const float* __restrict__ A, const float* __restrict__ Bt,
                                                  float* __restrict__ C, int M, int N, int K, float sa, float sb) {
    const int lane = threadIdx.x, l31 = lane & 31, half = lane >> 5;
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* a = A + (size_t)(m0 + l31) * K;        // A row-major [M][K]
    const float* b = Bt + (size_t)(n0 + l31) * K;       // B stored transposed [N][K]
    if (VAR == 0) {
        for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k + half], b[k + half], acc, 0, 0, 0);
    } else if (VAR >= 4) {
        for (int k = 0; k < K; k += 16) {
            float xa[8], xb[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { xa[j] = a[k + 8 * half + j]; xb[j] = b[k + 8 * half + j]; }
            f16x8 a1, a2, b1, b2;
            split2h(xa, sa, a1, a2);
            split2h(xb, sb, b1, b2);
            if (VAR == 5) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc, 0, 0, 0);
        }
        const float inv = 1.0f / (sa * sb);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] *= inv;
    } else {
        for (int k = 0; k < K; k += 16) {
            float xa[8], xb[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { xa[j] = a[k + 8 * half + j]; xb[j] = b[k + 8 * half + j]; }
            bf16x8 a1, a2, a3, b1, b2, b3;
            split3(xa, a1, a2, a3);
            split3(xb, b1, b2, b3);
            if (VAR == 3) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b3, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b2, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b3, acc, 0, 0, 0);
            }
            if (VAR == 1 || VAR == 3) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        C[(size_t)(m0 + row) * N + n0 + l31] = acc[r];
    }
}

static double urand(unsigned long long& s) {
    s = s * 6364136223846793005ULL + 1442695040888963407ULL;
    return (double)(s >> 11) / 9007199254740992.0;
}
static double nrand(unsigned long long& s) {
    const double u = urand(s) + 1e-300, v = urand(s);
    return std::sqrt(-2.0 * std::log(u)) * std::cos(6.283185307179586 * v);
}

int main() {
    const int M = 128, N = 128;
    const char* names[7] = {"fp32 mfma 32x32x2", "bf16x6", "bf16x3", "bf16x9", "f16x3 scaled", "f16x4 scaled", "f16x3 unscaled"};
    for (int dist = 0; dist < 5; ++dist)
    for (int K : {1728, 3456, 13824}) {
        std::vector<float> A((size_t)M * K), B((size_t)N * K);
        unsigned long long s = 1234 + K + dist;
        for (auto& v : A) {
            double x = nrand(s);
            if (dist == 0) x = x / (1.0 + std::exp(-x));                    // SiLU of a normal: conv inputs
            if (dist == 2) x = x * std::exp(4.0 * nrand(s));                 // wide dynamic range
            if (dist == 3) { if (urand(s) < 1e-3) x *= 60.0; x = x / (1.0 + std::exp(-x)); }   // SiLU with rare 60-sigma outliers
            if (dist == 4) { x = 0.02 * x; x = x / (1.0 + std::exp(-x)); }                      // tiny activations (|x| ~ 0.01)
            v = (float)x;
        }
        for (auto& v : B) v = (float)(nrand(s) / std::sqrt((double)K) * (dist == 2 ? std::exp(4.0 * nrand(s)) : 1.0));
        std::vector<double> ref((size_t)M * N);
        double scale = 0.0;
        for (int m = 0; m < M; ++m)
            for (int n = 0; n < N; ++n) {
                double acc = 0.0;
                for (int k = 0; k < K; ++k) acc += (double)A[(size_t)m * K + k] * (double)B[(size_t)n * K + k];
                ref[(size_t)m * N + n] = acc;
                scale = std::fmax(scale, std::fabs(acc));
            }
        float *dA, *dB, *dC;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, (size_t)M * N * 4);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        std::vector<float> C((size_t)M * N);
        printf("dist %d (%s) K=%d  max|C| %.3e\n", dist, dist == 0 ? "silu(normal) x normal" : dist == 1 ? "normal x normal" : dist == 2 ? "lognormal-scaled" : dist == 3 ? "silu(normal + 60-sigma outliers) x normal" : "silu(0.02 normal) x normal", K, scale);
        double bmax = 0.0;
        for (auto v : B) bmax = std::fmax(bmax, std::fabs((double)v));
        int eb; std::frexp(bmax, &eb);                       // bmax = f * 2^eb, f in [0.5, 1)
        const float sb = std::ldexp(1.0f, 15 - eb), sa = 8.0f;   // max|B| * sb in [2^14, 2^15)
        for (int var = 0; var < 7; ++var) {
            dim3 grid(M / 32, N / 32);
            if (var == 0) hipLaunchKernelGGL(gemm_kernel<0>, grid, dim3(64), 0, 0, dA, dB, dC, M, N, K, 1.f, 1.f);
            if (var == 1) hipLaunchKernelGGL(gemm_kernel<1>, grid, dim3(64), 0, 0, dA, dB, dC, M, N, K, 1.f, 1.f);
            if (var == 2) hipLaunchKernelGGL(gemm_kernel<2>, grid, dim3(64), 0, 0, dA, dB, dC, M, N, K, 1.f, 1.f);
            if (var == 3) hipLaunchKernelGGL(gemm_kernel<3>, grid, dim3(64), 0, 0, dA, dB, dC, M, N, K, 1.f, 1.f);
            if (var == 4) hipLaunchKernelGGL(gemm_kernel<4>, grid, dim3(64), 0, 0, dA, dB, dC, M, N, K, sa, sb);
            if (var == 5) hipLaunchKernelGGL(gemm_kernel<5>, grid, dim3(64), 0, 0, dA, dB, dC, M, N, K, sa, sb);
            if (var == 6) hipLaunchKernelGGL(gemm_kernel<4>, grid, dim3(64), 0, 0, dA, dB, dC, M, N, K, 1.f, 1.f);
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
            hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
            double mx = 0.0, sq = 0.0;
            for (size_t i = 0; i < C.size(); ++i) {
                const double e = std::fabs((double)C[i] - ref[i]);
                mx = std::fmax(mx, e); sq += e * e;
            }
            printf("   %-18s max err/scale %.3e   rms err/scale %.3e\n", names[var], mx / scale, std::sqrt(sq / C.size()) / scale);
        }
        hipFree(dA); hipFree(dB); hipFree(dC);
    }
    return 0;
}
