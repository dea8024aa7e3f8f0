// lpips.hip -- the pieces of the LPIPS (AlexNet, v0.1) perceptual distance that are not convolutions.
//
// The sender's decision rule compares generated frames with the originals by LPIPS (reference city_sender.py:302 builds
// lpips.LPIPS(net='alex'), :376-406 decide_5to5_lpips keeps a frame while the distance stays under the threshold).  The
// algorithm as the reference vendors it (models/networks_basic.py:62-93 PNetLin.forward + ScalingLayer, models/eval_models.py:35-37
// normalize_tensor, models/pretrained_networks.py:56-94 AlexNet slices; restated in oracle/lpips.py):
//     x -> (x - shift) / scale                                            (ScalingLayer, per RGB channel)
//     AlexNet features: conv 11x11 s4 p2 (3->64) ReLU | maxpool 3 s2, conv 5x5 p2 (64->192) ReLU |
//                       maxpool 3 s2, conv 3x3 (192->384) ReLU | conv 3x3 (384->256) ReLU | conv 3x3 (256->256) ReLU
//     per tap k of the five ReLU outputs: unit-normalise over channels (x / (|x|_2 + 1e-10)), squared difference of the two
//     images, 1x1 convolution with the learned non-negative weights lin_k (one output channel), spatial mean; sum over k.
// The convolutions run on evc_conv2d_nhwc_f32 (exact bf16 split: the inputs are not normalised); this file holds
//   * evc_im2col_nchw_f32: the stride-4 11x11 first convolution as patches -> rows of a matrix (the scaling layer fused in, zero
//     padding applied AFTER the scaling as the convolution does), so that it becomes a 1x1 convolution over 363 (padded 368) channels;
//   * evc_maxpool3s2_nhwc_f32: MaxPool2d(kernel 3, stride 2), floor mode, no padding;
//   * evc_lpips_layer_f32: normalise + squared difference + lin_k + spatial mean of one tap, accumulated into the distance.
#include <hip/hip_runtime.h>
#include "../../include/evc_hip.h"

namespace {

// out[n][oy][ox][(c*KH + ky)*KW + kx] = in-range ? (x[n][c][oy*s - pad + ky][ox*s - pad + kx] - shift[c]) / scale[c] : 0;
// channels K = C*KH*KW .. ld_out-1 are zero.  One thread per output element (coalesced writes along the patch axis).
__global__ void im2col_kernel(const float* __restrict__ x, float* __restrict__ out, int C, int H, int W, int KH, int KW,
                              int stride, int pad, int Ho, int Wo, int ld_out, const float* __restrict__ shift,
                              const float* __restrict__ scale, size_t total) {
    const int K = C * KH * KW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % ld_out);
        const size_t pix = i / ld_out;
        const int ox = (int)(pix % Wo);
        const int oy = (int)((pix / Wo) % Ho);
        const size_t n = pix / ((size_t)Wo * Ho);
        float v = 0.f;
        if (k < K) {
            const int c = k / (KH * KW), r = k - c * KH * KW;
            const int ky = r / KW, kx = r - ky * KW;
            const int yy = oy * stride - pad + ky, xx = ox * stride - pad + kx;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                v = x[((n * C + c) * H + yy) * W + xx];
                if (shift) v = (v - shift[c]) / scale[c];
            }
        }
        out[i] = v;
    }
}

__global__ void maxpool3s2_kernel(const float4* __restrict__ x, float4* __restrict__ out, int H, int W, int C4, int Ho, int Wo,
                                  size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        const size_t pix = i / C4;
        const int ox = (int)(pix % Wo);
        const int oy = (int)((pix / Wo) % Ho);
        const size_t n = pix / ((size_t)Wo * Ho);
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float4 v = x[((n * H + (2 * oy + ky)) * W + (2 * ox + kx)) * C4 + c];
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        out[i] = m;
    }
}

// One workgroup (256 threads = 4 waves) per image pair; a wave takes pixels wave, wave + 4, ...; its lanes stride the channels.
// Fixed assignment and fixed reduction trees: deterministic.
__global__ __launch_bounds__(256) void lpips_layer_kernel(const float* __restrict__ f0, const float* __restrict__ f1,
                                                          const float* __restrict__ w, float* __restrict__ dist, int HW, int C,
                                                          int accumulate) {
    __shared__ float red[4];
    const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* a = f0 + (size_t)n * HW * C;
    const float* b = f1 + (size_t)n * HW * C;
    float acc = 0.f;
    for (int p = wave; p < HW; p += 4) {
        const float* ap = a + (size_t)p * C;
        const float* bp = b + (size_t)p * C;
        float sa = 0.f, sb = 0.f;
        for (int c = lane; c < C; c += 64) { const float u = ap[c], v = bp[c]; sa += u * u; sb += v * v; }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { sa += __shfl_xor(sa, off); sb += __shfl_xor(sb, off); }
        const float ia = 1.0f / (sqrtf(sa) + 1e-10f), ib = 1.0f / (sqrtf(sb) + 1e-10f);
        float d = 0.f;
        for (int c = lane; c < C; c += 64) { const float t = ap[c] * ia - bp[c] * ib; d += w[c] * t * t; }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) d += __shfl_xor(d, off);
        acc += d;                                  // identical in every lane
    }
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float v = (red[0] + red[1] + red[2] + red[3]) / (float)HW;
        dist[n] = accumulate ? dist[n] + v : v;
    }
}

}  // namespace

extern "C" int evc_im2col_nchw_f32(const float* x, float* out, int N, int C, int H, int W, int KH, int KW, int stride, int pad,
                                   int ld_out, const float* shift, const float* scale, void* stream) {
    if (!x || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0) return EVC_EINVAL;
    if ((shift == nullptr) != (scale == nullptr)) return EVC_EINVAL;
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    if (Ho <= 0 || Wo <= 0 || ld_out < C * KH * KW) return EVC_EINVAL;
    const size_t total = (size_t)N * Ho * Wo * ld_out;
    const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(im2col_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, out, C, H, W, KH, KW, stride, pad, Ho, Wo,
                       ld_out, shift, scale, total);
    return hipGetLastError() == hipSuccess ? EVC_OK : EVC_ELAUNCH;
}

extern "C" int evc_maxpool3s2_nhwc_f32(const float* x, float* out, int N, int H, int W, int C, void* stream) {
    if (!x || !out || N <= 0 || H < 3 || W < 3 || C <= 0 || (C & 3)) return EVC_EINVAL;
    const int Ho = (H - 3) / 2 + 1, Wo = (W - 3) / 2 + 1;
    const size_t total = (size_t)N * Ho * Wo * (C / 4);
    const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(maxpool3s2_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(x),
                       reinterpret_cast<float4*>(out), H, W, C / 4, Ho, Wo, total);
    return hipGetLastError() == hipSuccess ? EVC_OK : EVC_ELAUNCH;
}

extern "C" int evc_lpips_layer_f32(const float* f0, const float* f1, const float* lin_w, float* dist, int N, int HW, int C,
                                   int accumulate, void* stream) {
    if (!f0 || !f1 || !lin_w || !dist || N <= 0 || HW <= 0 || C <= 0) return EVC_EINVAL;
    hipLaunchKernelGGL(lpips_layer_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, f0, f1, lin_w, dist, HW, C, accumulate);
    return hipGetLastError() == hipSuccess ? EVC_OK : EVC_ELAUNCH;
}
