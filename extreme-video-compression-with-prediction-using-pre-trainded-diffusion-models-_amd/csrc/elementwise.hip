// elementwise.hip -- layout changes at the NCHW boundary, sampler update steps, ELIC glue.
// All HBM-bound streaming kernels: one pass, grid-stride, float4 where the layout allows it.
#include <hip/hip_runtime.h>
#include "../../include/evc_hip.h"

namespace {

inline int grid_for(size_t n, int cap = 4096) {
    size_t g = (n + 255) / 256;
    return (int)(g > (size_t)cap ? (size_t)cap : (g ? g : 1));
}
#define EVC_LAUNCH_OK() (hipGetLastError() == hipSuccess ? EVC_OK : EVC_ELAUNCH)

// out[b][h][w][c] = c < C0 ? x0[b][c][h][w] : c < C0+C1 ? x1[b][c-C0][h][w] : 0.
// One thread = one pixel, all of its Cpad channels: every plane is read coalesced along w (lanes walk pixels) and the
// pixel's row is written as float4s, so a wave writes 64 * Cpad * 4 contiguous bytes (round 1 wrote one float per lane
// at a Cpad * 4-byte stride: 1.2 TB/s).
__global__ void pack_nchw_to_nhwc_kernel(const float* __restrict__ x0, int C0, const float* __restrict__ x1, int C1,
                                         float* __restrict__ out, int Cpad, int HW, size_t npix) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / HW;
        const size_t p = i - b * HW;
        const float* s0 = x0 + b * C0 * HW + p;
        const float* s1 = C1 > 0 ? x1 + b * C1 * HW + p : x0;
        const auto ld = [&](int c) { return c < C0 ? s0[(size_t)c * HW] : (c < C0 + C1 ? s1[(size_t)(c - C0) * HW] : 0.f); };
        float* o = out + i * Cpad;
        int c = 0;
        if ((Cpad & 3) == 0)
            for (; c + 4 <= Cpad; c += 4) *reinterpret_cast<float4*>(o + c) = make_float4(ld(c), ld(c + 1), ld(c + 2), ld(c + 3));
        for (; c < Cpad; ++c) o[c] = ld(c);
    }
}

// out[b][c][h][w] = in[b][h][w][c] (row stride ld): one thread = one pixel, its C values read as one contiguous row,
// every output plane written coalesced along w.
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ in, int ld, float* __restrict__ out, int C, int HW,
                                    size_t npix) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / HW;
        const size_t p = i - b * HW;
        const float* r = in + i * ld;
        float* o = out + b * C * HW + p;
        int c = 0;
        if ((ld & 3) == 0)
            for (; c + 4 <= C; c += 4) {
                const float4 v = *reinterpret_cast<const float4*>(r + c);
                o[(size_t)c * HW] = v.x; o[(size_t)(c + 1) * HW] = v.y; o[(size_t)(c + 2) * HW] = v.z; o[(size_t)(c + 3) * HW] = v.w;
            }
        for (; c < C; ++c) o[(size_t)c * HW] = r[c];
    }
}

__global__ void ddpm_step_kernel(float* __restrict__ x, const float* __restrict__ e, const float* __restrict__ noise,
                                 size_t n, float k1, float k2, float c1, float c2, float sigma, int clip) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float xv = x[i];
        float x0 = k1 * (xv - k2 * e[i]);
        if (clip) x0 = fminf(fmaxf(x0, -1.f), 1.f);
        float y = c1 * x0 + c2 * xv;
        if (noise) y += sigma * noise[i];
        x[i] = y;
    }
}

__global__ void ddim_step_kernel(float* __restrict__ x, const float* __restrict__ e, size_t n, float k1, float k2,
                                 float c1, float c2, int clip) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float ev = e[i];
        float x0 = k1 * (x[i] - k2 * ev);
        if (clip) x0 = fminf(fmaxf(x0, -1.f), 1.f);
        x[i] = c1 * x0 + c2 * ev;
    }
}

__global__ void axpy_kernel(const float* __restrict__ x, const float* __restrict__ e, float* __restrict__ y, size_t n,
                            float alpha) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = x[i] + alpha * e[i];
}

__global__ void pndm_transfer_kernel(const float* __restrict__ x, const float* __restrict__ e, float* __restrict__ y,
                                     size_t n, float d, float cx, float ce, int clip) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float xv = x[i];
        float v = xv + d * (cx * xv - ce * e[i]);
        if (clip) v = fminf(fmaxf(v, -1.f), 1.f);
        y[i] = v;
    }
}

__global__ void lincomb4_kernel(const float* __restrict__ e0, const float* __restrict__ e1,
                                const float* __restrict__ e2, const float* __restrict__ e3, float* __restrict__ y,
                                size_t n, float w0, float w1, float w2, float w3) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float v = w0 * e0[i];
        if (e1) v += w1 * e1[i];
        if (e2) v += w2 * e2[i];
        if (e3) v += w3 * e3[i];
        y[i] = v;
    }
}

__global__ void scale_clamp_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n, float mul, float add,
                                   int clamp, float lo, float hi) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float v = x[i] * mul + add;
        if (clamp) v = fminf(fmaxf(v, lo), hi);
        y[i] = v;
    }
}

__global__ void gate_residual_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                     const float* __restrict__ x, float* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float s = 1.0f / (1.0f + expf(-b[i]));
        out[i] = a[i] * s + x[i];
    }
}

// Checkerboard site of packed column j in row h: anchors (parity 0) sit at w = 2j + (h & 1),
// non-anchors (parity 1) at w = 2j + 1 - (h & 1)   (reference Network.py:488-491, 507-510).
__device__ __forceinline__ int cb_col(int h, int j, int parity) { return 2 * j + ((h & 1) ^ parity); }

__global__ void elic_gather_params_kernel(const float* __restrict__ ms, int ld, int mean_off, int scale_off, int C,
                                          int H, int W, int parity, const float* __restrict__ table, int n_scales,
                                          int* __restrict__ idx, float* __restrict__ means, size_t total) {
    const int Wh = W >> 1;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % Wh);
        size_t r = i / Wh;
        const int h = (int)(r % H); r /= H;
        const int c = (int)(r % C);
        const int b = (int)(r / C);
        const int w = cb_col(h, j, parity);
        const float* px = ms + ((size_t)(b * H + h) * W + w) * ld;
        // GaussianConditional.build_indexes (compressai 1.1.5): scales = max(scales, 0.11);
        // index = (n-1) - #{s in table[:-1] : scales <= s}
        const float sc = fmaxf(px[scale_off + c], 0.11f);
        int id = n_scales - 1;
        for (int t = 0; t < n_scales - 1; ++t) id -= (sc <= table[t]) ? 1 : 0;
        idx[i] = id;
        means[i] = px[mean_off + c];
    }
}

__global__ void elic_scatter_symbols_kernel(const int* __restrict__ sym, const float* __restrict__ means,
                                            float* __restrict__ y_hat, int ld, int c0, int C, int H, int W,
                                            int parity, size_t total) {
    const int Wh = W >> 1;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % Wh);
        size_t r = i / Wh;
        const int h = (int)(r % H); r /= H;
        const int c = (int)(r % C);
        const int b = (int)(r / C);
        const int w = cb_col(h, j, parity);
        y_hat[((size_t)(b * H + h) * W + w) * ld + c0 + c] = (float)sym[i] + means[i];
    }
}

__global__ void elic_quantize_kernel(const float* __restrict__ y, int ld, int c0, const float* __restrict__ means,
                                     int C, int H, int W, int parity, int* __restrict__ sym, size_t total) {
    const int Wh = W >> 1;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % Wh);
        size_t r = i / Wh;
        const int h = (int)(r % H); r /= H;
        const int c = (int)(r % C);
        const int b = (int)(r / C);
        const int w = cb_col(h, j, parity);
        sym[i] = (int)rintf(y[((size_t)(b * H + h) * W + w) * ld + c0 + c] - means[i]);   // round half to even
    }
}

}  // namespace

extern "C" int evc_pack_nchw_to_nhwc_f32(const float* x0, int C0, const float* x1, int C1, float* out, int Cpad,
                                         int B, int H, int W, void* stream) {
    if (!x0 || !out || C0 <= 0 || C1 < 0 || (C1 > 0 && !x1) || Cpad < C0 + C1 || B <= 0 || H <= 0 || W <= 0)
        return EVC_EINVAL;
    const size_t npix = (size_t)B * H * W;
    hipLaunchKernelGGL(pack_nchw_to_nhwc_kernel, dim3(grid_for(npix)), dim3(256), 0, (hipStream_t)stream, x0, C0, x1,
                       C1, out, Cpad, H * W, npix);
    return EVC_LAUNCH_OK();
}

extern "C" int evc_nhwc_to_nchw_f32(const float* in, int ld, float* out, int B, int C, int H, int W, void* stream) {
    if (!in || !out || C <= 0 || ld < C || B <= 0 || H <= 0 || W <= 0) return EVC_EINVAL;
    const size_t npix = (size_t)B * H * W;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for(npix)), dim3(256), 0, (hipStream_t)stream, in, ld, out, C,
                       H * W, npix);
    return EVC_LAUNCH_OK();
}

extern "C" int evc_ddpm_step_f32(float* x, const float* e, const float* noise, long long n, float k1, float k2,
                                 float c1, float c2, float sigma, int clip, void* stream) {
    if (!x || !e || n <= 0) return EVC_EINVAL;
    hipLaunchKernelGGL(ddpm_step_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, e, noise, (size_t)n,
                       k1, k2, c1, c2, sigma, clip);
    return EVC_LAUNCH_OK();
}

extern "C" int evc_ddim_step_f32(float* x, const float* e, long long n, float k1, float k2, float c1, float c2,
                                 int clip, void* stream) {
    if (!x || !e || n <= 0) return EVC_EINVAL;
    hipLaunchKernelGGL(ddim_step_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, e, (size_t)n, k1,
                       k2, c1, c2, clip);
    return EVC_LAUNCH_OK();
}

extern "C" int evc_axpy_f32(const float* x, const float* e, float* y, long long n, float alpha, void* stream) {
    if (!x || !e || !y || n <= 0) return EVC_EINVAL;
    hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, e, y, (size_t)n, alpha);
    return EVC_LAUNCH_OK();
}

extern "C" int evc_pndm_transfer_f32(const float* x, const float* e, float* y, long long n, float d, float cx,
                                     float ce, int clip, void* stream) {
    if (!x || !e || !y || n <= 0) return EVC_EINVAL;
    hipLaunchKernelGGL(pndm_transfer_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, e, y, (size_t)n,
                       d, cx, ce, clip);
    return EVC_LAUNCH_OK();
}

extern "C" int evc_lincomb4_f32(const float* e0, const float* e1, const float* e2, const float* e3, float* y,
                                long long n, float w0, float w1, float w2, float w3, void* stream) {
    if (!e0 || !y || n <= 0) return EVC_EINVAL;
    hipLaunchKernelGGL(lincomb4_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, e0, e1, e2, e3, y,
                       (size_t)n, w0, w1, w2, w3);
    return EVC_LAUNCH_OK();
}

extern "C" int evc_scale_clamp_f32(const float* x, float* y, long long n, float mul, float add, int clamp, float lo,
                                   float hi, void* stream) {
    if (!x || !y || n <= 0) return EVC_EINVAL;
    hipLaunchKernelGGL(scale_clamp_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, (size_t)n, mul,
                       add, clamp, lo, hi);
    return EVC_LAUNCH_OK();
}

extern "C" int evc_gate_residual_f32(const float* a, const float* b, const float* x, float* out, long long n,
                                     void* stream) {
    if (!a || !b || !x || !out || n <= 0) return EVC_EINVAL;
    hipLaunchKernelGGL(gate_residual_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, x, out,
                       (size_t)n);
    return EVC_LAUNCH_OK();
}

extern "C" int evc_elic_gather_params_f32(const float* ms, int ld, int mean_off, int scale_off, int C, int B, int H,
                                          int W, int parity, const float* scale_table, int n_scales, int* idx,
                                          float* means, void* stream) {
    if (!ms || !scale_table || !idx || !means || C <= 0 || B <= 0 || H <= 0 || W <= 0 || (W & 1) || n_scales < 2 ||
        parity < 0 || parity > 1 || mean_off < 0 || scale_off < 0 || mean_off + C > ld || scale_off + C > ld)
        return EVC_EINVAL;
    const size_t total = (size_t)B * C * H * (W >> 1);
    hipLaunchKernelGGL(elic_gather_params_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, ms, ld,
                       mean_off, scale_off, C, H, W, parity, scale_table, n_scales, idx, means, total);
    return EVC_LAUNCH_OK();
}

extern "C" int evc_elic_scatter_symbols_f32(const int* symbols, const float* means, float* y_hat, int ld, int c0,
                                            int C, int B, int H, int W, int parity, void* stream) {
    if (!symbols || !means || !y_hat || C <= 0 || c0 < 0 || c0 + C > ld || B <= 0 || H <= 0 || W <= 0 || (W & 1) ||
        parity < 0 || parity > 1)
        return EVC_EINVAL;
    const size_t total = (size_t)B * C * H * (W >> 1);
    hipLaunchKernelGGL(elic_scatter_symbols_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, symbols,
                       means, y_hat, ld, c0, C, H, W, parity, total);
    return EVC_LAUNCH_OK();
}

extern "C" int evc_elic_quantize_f32(const float* y, int ld, int c0, const float* means, int C, int B, int H, int W,
                                     int parity, int* symbols, void* stream) {
    if (!y || !means || !symbols || C <= 0 || c0 < 0 || c0 + C > ld || B <= 0 || H <= 0 || W <= 0 || (W & 1) ||
        parity < 0 || parity > 1)
        return EVC_EINVAL;
    const size_t total = (size_t)B * C * H * (W >> 1);
    hipLaunchKernelGGL(elic_quantize_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, y, ld, c0,
                       means, C, H, W, parity, symbols, total);
    return EVC_LAUNCH_OK();
}
