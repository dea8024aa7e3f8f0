// frames.hip -- the three operations of the pseudo-3-D score network (config.model.arch = unetmorepseudo3d) that act ALONG
// THE FRAME AXIS of a video activation and have no 2-D counterpart in the other kernels:
//
//   * frame_group_norm : nn.GroupNorm of AttnBlockpp1d on (B*H*W, C, N): moments over (C / G channels x N frames) of ONE
//                        pixel, affine (reference models/better/layers3d.py:89-90,107);
//   * frame_attention  : the softmax attention of AttnBlockpp1d over the N frames of one pixel (layers3d.py:112-118);
//   * frame_mix        : the 1x1 "converter" convolutions over the frame axis, N -> M frames
//                        (reference models/better/ncsnpp_more.py:213-216,226-228,328-335,344-351);
//   * frame_taps       : the temporal taps of nn.Conv3d (arch = unetmore3d, layers3d.py:225-243) laid side by side along
//                        the channels -- frames n - 1 | n | n + 1, zeros beyond a sample's first / last frame -- so that the
//                        3 x 3 x 3 convolution is ONE 3 x 3 convolution with 3C input channels.
//
// Everything else of that network runs on the 2-D kernels: activations are kept frame-major inside a sample,
// x[b][n][pixel][c] = an NHWC tensor of B*N images, so a per-frame Conv2d is an ordinary convolution over B*N images, the
// Conv1d over the frames (PseudoConv3d.time_conv, layers3d.py:274,294-297) is a KH x 1 convolution over an "image" of
// N rows x (H*W) columns, and the 3-D GroupNorm's moments are the per-frame moments read as N times as many pixel runs.
//
// All four are HBM-bound element / pixel-wise passes (N <= 8 frames): one read + one write of the tensor.
#include <hip/hip_runtime.h>
#include "../../include/evc_hip.h"

namespace {

constexpr int MAX_FRAMES = 8;

// grid = B * HW pixels, block 256.  LDS: [2][C] channel moments + [2][G] group mean / rstd.
__global__ __launch_bounds__(256) void frame_group_norm_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, int N, int HW, int C, int G,
                                                               float eps) {
    extern __shared__ float lds[];
    float* s1 = lds;
    float* s2 = lds + C;
    float* gm = lds + 2 * C;
    float* gr = gm + G;
    const int b = blockIdx.x / HW, p = blockIdx.x - b * HW;
    const size_t frame = (size_t)HW * C;
    const float* xp = x + ((size_t)b * N * HW + p) * C;
    float* yp = y + ((size_t)b * N * HW + p) * C;
    for (int c = threadIdx.x; c < C; c += 256) {
        float a = 0.f, q = 0.f;
        for (int n = 0; n < N; ++n) {
            const float v = xp[n * frame + c];
            a += v;
            q += v * v;
        }
        s1[c] = a;
        s2[c] = q;
    }
    __syncthreads();
    const int cg = C / G;
    for (int g = threadIdx.x; g < G; g += 256) {
        float a = 0.f, q = 0.f;
        for (int j = 0; j < cg; ++j) {
            a += s1[g * cg + j];
            q += s2[g * cg + j];
        }
        const float inv = 1.0f / (float)(cg * N);
        const float mean = a * inv;
        const float var = fmaxf(q * inv - mean * mean, 0.f);
        gm[g] = mean;
        gr[g] = rsqrtf(var + eps);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        const int g = c / cg;
        const float a = gr[g] * gamma[c];
        const float s = beta[c] - gm[g] * a;
        for (int n = 0; n < N; ++n) yp[n * frame + c] = xp[n * frame + c] * a + s;
    }
}

// grid = (B * HW pixels, heads), block 64 (one wave).  Lane l owns channels l, l + 64, ... of the head; the N x N scores
// are per-lane partial dot products reduced over the wave.
__global__ __launch_bounds__(64) void frame_attention_kernel(const float* __restrict__ qkv, int ld, float* __restrict__ out,
                                                             int ld_out, int N, int HW, int C, int D, float scale) {
    const int b = blockIdx.x / HW, p = blockIdx.x - b * HW;
    const int c0 = blockIdx.y * D;
    const size_t frame = (size_t)HW * ld;
    const float* base = qkv + ((size_t)b * N * HW + p) * ld + c0;
    float s[MAX_FRAMES][MAX_FRAMES];
#pragma unroll
    for (int t = 0; t < MAX_FRAMES; ++t)
#pragma unroll
        for (int i = 0; i < MAX_FRAMES; ++i) s[t][i] = 0.f;
    for (int c = threadIdx.x; c < D; c += 64) {
        float q[MAX_FRAMES], k[MAX_FRAMES];
#pragma unroll
        for (int n = 0; n < MAX_FRAMES; ++n) {
            q[n] = n < N ? base[n * frame + c] : 0.f;
            k[n] = n < N ? base[n * frame + C + c] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < MAX_FRAMES; ++t)
#pragma unroll
            for (int i = 0; i < MAX_FRAMES; ++i) s[t][i] = fmaf(q[t], k[i], s[t][i]);
    }
#pragma unroll
    for (int t = 0; t < MAX_FRAMES; ++t)
#pragma unroll
        for (int i = 0; i < MAX_FRAMES; ++i) {
            float v = s[t][i];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            s[t][i] = v * scale;
        }
    // softmax over the keys i < N of every query row t
#pragma unroll
    for (int t = 0; t < MAX_FRAMES; ++t) {
        float m = -INFINITY;
#pragma unroll
        for (int i = 0; i < MAX_FRAMES; ++i)
            if (i < N) m = fmaxf(m, s[t][i]);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < MAX_FRAMES; ++i) {
            s[t][i] = i < N ? expf(s[t][i] - m) : 0.f;
            sum += s[t][i];
        }
        const float inv = 1.0f / sum;
#pragma unroll
        for (int i = 0; i < MAX_FRAMES; ++i) s[t][i] *= inv;
    }
    float* obase = out + ((size_t)b * N * HW + p) * ld_out + c0;
    const size_t oframe = (size_t)HW * ld_out;
    for (int c = threadIdx.x; c < D; c += 64) {
        float v[MAX_FRAMES];
#pragma unroll
        for (int n = 0; n < MAX_FRAMES; ++n) v[n] = n < N ? base[n * frame + 2 * C + c] : 0.f;
#pragma unroll
        for (int t = 0; t < MAX_FRAMES; ++t) {
            if (t < N) {
                float o = 0.f;
#pragma unroll
                for (int i = 0; i < MAX_FRAMES; ++i) o = fmaf(s[t][i], v[i], o);
                obase[t * oframe + c] = o;
            }
        }
    }
}

struct MixW {
    float w[MAX_FRAMES * MAX_FRAMES];
    float b[MAX_FRAMES];
};

// y[b][m][j] = sum_n w[m][n] x[b][n][j] + bias[m]; float4 along j.
__global__ __launch_bounds__(256) void frame_mix_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                        const float* __restrict__ w, const float* __restrict__ bias, int B,
                                                        int N, int M, size_t inner4) {
    __shared__ MixW mw;
    if (threadIdx.x < N * M) mw.w[threadIdx.x] = w[threadIdx.x];
    if (threadIdx.x < M) mw.b[threadIdx.x] = bias[threadIdx.x];
    __syncthreads();
    const size_t total = (size_t)B * inner4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t b = i / inner4, j = i - b * inner4;
        const float4* xs = reinterpret_cast<const float4*>(x) + b * N * inner4 + j;
        float4* ys = reinterpret_cast<float4*>(y) + b * M * inner4 + j;
        float4 v[MAX_FRAMES];
#pragma unroll
        for (int n = 0; n < MAX_FRAMES; ++n) v[n] = n < N ? xs[n * inner4] : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int m = 0; m < M; ++m) {
            float4 o = make_float4(mw.b[m], mw.b[m], mw.b[m], mw.b[m]);
#pragma unroll
            for (int n = 0; n < MAX_FRAMES; ++n) {
                const float wn = n < N ? mw.w[m * N + n] : 0.f;
                o.x = fmaf(wn, v[n].x, o.x); o.y = fmaf(wn, v[n].y, o.y);
                o.z = fmaf(wn, v[n].z, o.z); o.w = fmaf(wn, v[n].w, o.w);
            }
            ys[m * inner4] = o;
        }
    }
}

// y[b][n][p][kt * C + c] = x[b][n + kt - 1][p][c] (0 outside the sample's frames), kt = 0, 1, 2; float4 along c.
__global__ __launch_bounds__(256) void frame_taps_kernel(const float* __restrict__ x, float* __restrict__ y, int N, size_t HW,
                                                         int C4, size_t total) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        size_t r = i / C4;
        const int kt = (int)(r % 3);
        r /= 3;                                   // r = (b * N + n) * HW + p
        const size_t p = r % HW, bn = r / HW;
        const int n = (int)(bn % N) + kt - 1;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n >= 0 && n < N) v = reinterpret_cast<const float4*>(x)[((bn + kt - 1) * HW + p) * C4 + c4];
        reinterpret_cast<float4*>(y)[i] = v;
    }
}

}  // namespace

extern "C" int evc_frame_taps_f32(const float* x, float* y, int B, int N, int HW, int C, void* stream) {
    if (!x || !y || B <= 0 || N <= 0 || HW <= 0 || C <= 0 || (C & 3)) return EVC_EINVAL;
    const size_t total = (size_t)B * N * HW * 3 * (C >> 2);
    const int grid = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
    hipLaunchKernelGGL(frame_taps_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, y, N, (size_t)HW, C >> 2, total);
    return hipGetLastError() == hipSuccess ? EVC_OK : EVC_ELAUNCH;
}

extern "C" int evc_frame_group_norm_f32(const float* x, float* y, const float* gamma, const float* beta, int B, int N,
                                        int HW, int C, int groups, float eps, void* stream) {
    if (!x || !y || !gamma || !beta || B <= 0 || N <= 0 || HW <= 0 || C <= 0 || groups <= 0 || C % groups) return EVC_EINVAL;
    if ((long long)B * HW > 0x7fffffffLL) return EVC_EUNSUPPORTED;
    const size_t lds = (size_t)(2 * C + 2 * groups) * sizeof(float);
    if (lds > 64 * 1024) return EVC_EUNSUPPORTED;
    hipLaunchKernelGGL(frame_group_norm_kernel, dim3(B * HW), dim3(256), lds, (hipStream_t)stream, x, y, gamma, beta, N, HW,
                       C, groups, eps);
    return hipGetLastError() == hipSuccess ? EVC_OK : EVC_ELAUNCH;
}

extern "C" int evc_frame_attention_f32(const float* qkv, int ld_qkv, float* out, int ld_out, int B, int N, int HW, int C,
                                       int heads, float scale, void* stream) {
    if (!qkv || !out || B <= 0 || N <= 0 || HW <= 0 || C <= 0 || heads <= 0 || C % heads || ld_qkv < 3 * C || ld_out < C)
        return EVC_EINVAL;
    if (N > MAX_FRAMES || heads > 65535 || (long long)B * HW > 0x7fffffffLL) return EVC_EUNSUPPORTED;
    hipLaunchKernelGGL(frame_attention_kernel, dim3(B * HW, heads), dim3(64), 0, (hipStream_t)stream, qkv, ld_qkv, out,
                       ld_out, N, HW, C, C / heads, scale);
    return hipGetLastError() == hipSuccess ? EVC_OK : EVC_ELAUNCH;
}

extern "C" int evc_frame_mix_f32(const float* x, float* y, const float* w, const float* bias, int B, int N, int M,
                                 long long inner, void* stream) {
    if (!x || !y || !w || !bias || B <= 0 || N <= 0 || M <= 0 || inner <= 0 || (inner & 3)) return EVC_EINVAL;
    if (N > MAX_FRAMES || M > MAX_FRAMES) return EVC_EUNSUPPORTED;
    const size_t inner4 = (size_t)inner / 4, total = (size_t)B * inner4;
    const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(frame_mix_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, y, w, bias, B, N, M, inner4);
    return hipGetLastError() == hipSuccess ? EVC_OK : EVC_ELAUNCH;
}
