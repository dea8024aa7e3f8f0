// fir.hip -- upfirdn2d: zero-insert upsample, pad, 2-D FIR (true convolution = flipped kernel), decimate.
//
// evc_upfirdn2d_f32      : drop-in for the reference's native op on NCHW planes
//                          (reference models/better/op/upfirdn2d_kernel.cu:107-243, upfirdn2d.py:163-204).
// evc_upfirdn2d_nhwc_f32 : same maths on NHWC with the AdaGN affine + SiLU of the producing norm applied
//                          to each input sample on load (reference models/better/layerspp.py:596-611 runs
//                          norm -> act -> resample as three passes).
// Both are HBM-bound gathers: only the non-zero taps of the zero-stuffed signal are visited
// ((kh/up)*(kw/up) loads per output), taps live in constant kernel arguments, NHWC moves float4 per lane.
#include <hip/hip_runtime.h>
#include "../../include/evc_hip.h"

namespace {

struct Taps { float k[64]; };   // flipped kernel: kf[a][b] = kernel[kh-1-a][kw-1-b]

struct FirGeom {
    int in_h, in_w, out_h, out_w, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_y0;
};

__device__ __forceinline__ float act_fn(float v, int act) {
    if (act == EVC_ACT_SILU) return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
    if (act == EVC_ACT_RELU) return fmaxf(v, 0.0f);
    return v;
}

// out(oy, ox) = sum_{a,b} kf[a][b] * z(oy*down + a - pad0, ox*down + b - pad0),
// z(Y, X) = in(Y/up, X/up) when Y, X are multiples of up and inside the image, else 0.
__global__ void upfirdn2d_nchw_kernel(const float* __restrict__ in, float* __restrict__ out, Taps t, FirGeom g,
                                      size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % g.out_w);
        size_t r = i / g.out_w;
        const int oy = (int)(r % g.out_h);
        const size_t plane = r / g.out_h;
        const float* ip = in + plane * (size_t)g.in_h * g.in_w;
        const int by = oy * g.down_y - g.pad_y0, bx = ox * g.down_x - g.pad_x0;
        float acc = 0.f;
        for (int a = 0; a < g.kh; ++a) {
            const int Y = by + a;
            if (Y < 0 || Y % g.up_y != 0) continue;
            const int iy = Y / g.up_y;
            if (iy >= g.in_h) continue;
            for (int b = 0; b < g.kw; ++b) {
                const int X = bx + b;
                if (X < 0 || X % g.up_x != 0) continue;
                const int ix = X / g.up_x;
                if (ix >= g.in_w) continue;
                acc += t.k[a * g.kw + b] * ip[(size_t)iy * g.in_w + ix];
            }
        }
        out[i] = acc;
    }
}

__global__ void upfirdn2d_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, Taps t, FirGeom g,
                                      int C, const float* __restrict__ ca, const float* __restrict__ cs, int act,
                                      size_t total4) {
    const int C4 = C >> 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        size_t r = i / C4;
        const int ox = (int)(r % g.out_w); r /= g.out_w;
        const int oy = (int)(r % g.out_h);
        const int b = (int)(r / g.out_h);
        const float* ip = in + (size_t)b * g.in_h * g.in_w * C + 4 * c4;
        float4 a4 = make_float4(1.f, 1.f, 1.f, 1.f), s4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ca) {
            a4 = *reinterpret_cast<const float4*>(ca + (size_t)b * C + 4 * c4);
            s4 = *reinterpret_cast<const float4*>(cs + (size_t)b * C + 4 * c4);
        }
        const int by = oy * g.down_y - g.pad_y0, bx = ox * g.down_x - g.pad_x0;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int a = 0; a < g.kh; ++a) {
            const int Y = by + a;
            if (Y < 0 || Y % g.up_y != 0) continue;
            const int iy = Y / g.up_y;
            if (iy >= g.in_h) continue;
            for (int bb = 0; bb < g.kw; ++bb) {
                const int X = bx + bb;
                if (X < 0 || X % g.up_x != 0) continue;
                const int ix = X / g.up_x;
                if (ix >= g.in_w) continue;
                float4 v = *reinterpret_cast<const float4*>(ip + ((size_t)iy * g.in_w + ix) * C);
                v.x = act_fn(v.x * a4.x + s4.x, act); v.y = act_fn(v.y * a4.y + s4.y, act);
                v.z = act_fn(v.z * a4.z + s4.z, act); v.w = act_fn(v.w * a4.w + s4.w, act);
                const float w = t.k[a * g.kw + bb];
                acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
            }
        }
        reinterpret_cast<float4*>(out)[i] = acc;
    }
}

// ---- the two configurations the score network uses (4x4 taps; models/better/up_or_down_sampling.py:196-258) ----------
// The generic gather above walks kh*kw taps with a divisibility test each and re-applies the AdaGN affine + SiLU once
// per tap per output: 16x per input element on the up path, 4x on the down path (measured 1.0 / 2.2 TB/s).  These two
// kernels unroll the taps at compile time and give every thread a small BLOCK of outputs so that each input element
// is loaded and activated once (up) / 2.25 times (down).

struct Taps16 { float k[4][4]; };    // flipped 4x4 kernel

template <int ACT>
__device__ __forceinline__ float4 fir_load(const float* __restrict__ p, bool ok, const float4& a, const float4& s,
                                           bool has_coef) {
    // branch-free: the caller clamps the address of an out-of-image sample to a legal one, the value is discarded by a
    // select AFTER the activation (zeros of the ACTIVATED tensor), so all loads of a thread can be in flight at once
    float4 v = *reinterpret_cast<const float4*>(p);
    if (has_coef) { v.x = v.x * a.x + s.x; v.y = v.y * a.y + s.y; v.z = v.z * a.z + s.z; v.w = v.w * a.w + s.w; }
    v.x = act_fn(v.x, ACT); v.y = act_fn(v.y, ACT); v.z = act_fn(v.z, ACT); v.w = act_fn(v.w, ACT);
    v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
    return v;
}

__device__ __forceinline__ void fma4(float4& acc, float w, const float4& v) {
    acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
}

// up = 2, pad (2, 1): output row 2i uses input rows i-1 (tap 0) and i (tap 2); row 2i+1 uses i (tap 1) and i+1
// (tap 3); same along x.  A thread owns 2x2 input cells = 4x4 outputs and reads the 4x4 input patch around them.
template <int ACT>
__global__ __launch_bounds__(256) void fir_up2_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                           Taps16 t, int B, int H, int W, int C,
                                                           const float* __restrict__ ca, const float* __restrict__ cs) {
    const int C4 = C >> 2, HB = H >> 1, WB = W >> 1;
    const size_t total = (size_t)B * HB * WB * C4;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % C4);
        size_t r = idx / C4;
        const int J = (int)(r % WB); r /= WB;
        const int I = (int)(r % HB);
        const int b = (int)(r / HB);
        const int i0 = 2 * I, j0 = 2 * J;
        float4 a4 = make_float4(1.f, 1.f, 1.f, 1.f), s4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ca) {
            a4 = *reinterpret_cast<const float4*>(ca + (size_t)b * C + 4 * c4);
            s4 = *reinterpret_cast<const float4*>(cs + (size_t)b * C + 4 * c4);
        }
        const float* ip = in + (size_t)b * H * W * C + 4 * c4;
        float4 v[4][4];                       // v[r][c] = act(in(i0 - 1 + r, j0 - 1 + c))
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const int y = i0 - 1 + rr, x = j0 - 1 + cc;
                const bool ok = y >= 0 && y < H && x >= 0 && x < W;
                v[rr][cc] = fir_load<ACT>(ip + ((size_t)(ok ? y : 0) * W + (ok ? x : 0)) * C, ok, a4, s4, ca != nullptr);
            }
        float* op = out + (size_t)b * (2 * H) * (2 * W) * C + 4 * c4;
#pragma unroll
        for (int di = 0; di < 2; ++di)
#pragma unroll
            for (int py = 0; py < 2; ++py)
#pragma unroll
                for (int dj = 0; dj < 2; ++dj)
#pragma unroll
                    for (int px = 0; px < 2; ++px) {
                        // input cell (i0+di, j0+dj) is patch position (1+di, 1+dj); phase 0: taps {0, 2} at offsets {-1, 0},
                        // phase 1: taps {1, 3} at offsets {0, +1}
                        const int ry = 1 + di + (py ? 0 : -1), rx = 1 + dj + (px ? 0 : -1);
                        const int ay = py ? 1 : 0, ax = px ? 1 : 0;
                        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                        fma4(acc, t.k[ay][ax], v[ry][rx]);
                        fma4(acc, t.k[ay][ax + 2], v[ry][rx + 1]);
                        fma4(acc, t.k[ay + 2][ax], v[ry + 1][rx]);
                        fma4(acc, t.k[ay + 2][ax + 2], v[ry + 1][rx + 1]);
                        const int oy = 2 * (i0 + di) + py, ox = 2 * (j0 + dj) + px;
                        *reinterpret_cast<float4*>(op + ((size_t)oy * (2 * W) + ox) * C) = acc;
                    }
    }
}

// down = 2, pad (1, 1): out(oy, ox) = sum_{a,b} kf[a][b] * in(2oy - 1 + a, 2ox - 1 + b).  A thread owns 2x2 outputs and
// streams the 6 input rows they touch, 6 samples each (36 loads for 4 outputs instead of 64).
template <int ACT>
__global__ __launch_bounds__(256) void fir_down2_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                             Taps16 t, int B, int H, int W, int C,
                                                             const float* __restrict__ ca, const float* __restrict__ cs) {
    const int C4 = C >> 2, OH = H >> 1, OW = W >> 1, HB = OH >> 1, WB = OW >> 1;
    const size_t total = (size_t)B * HB * WB * C4;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % C4);
        size_t r = idx / C4;
        const int J = (int)(r % WB); r /= WB;
        const int I = (int)(r % HB);
        const int b = (int)(r / HB);
        const int oy0 = 2 * I, ox0 = 2 * J;
        float4 a4 = make_float4(1.f, 1.f, 1.f, 1.f), s4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ca) {
            a4 = *reinterpret_cast<const float4*>(ca + (size_t)b * C + 4 * c4);
            s4 = *reinterpret_cast<const float4*>(cs + (size_t)b * C + 4 * c4);
        }
        const float* ip = in + (size_t)b * H * W * C + 4 * c4;
        float4 acc[2][2];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = 0; q < 2; ++q) acc[p][q] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int x0 = 2 * ox0 - 1, y0 = 2 * oy0 - 1;
#pragma unroll
        for (int rr = 0; rr < 6; ++rr) {
            const int y = y0 + rr;
            float4 v[6];
#pragma unroll
            for (int cc = 0; cc < 6; ++cc) {
                const int x = x0 + cc;
                const bool ok = y >= 0 && y < H && x >= 0 && x < W;
                v[cc] = fir_load<ACT>(ip + ((size_t)(ok ? y : 0) * W + (ok ? x : 0)) * C, ok, a4, s4, ca != nullptr);
            }
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int a = rr - 2 * p;                  // tap row of this input row for output row oy0 + p
                if (a < 0 || a > 3) continue;              // compile-time after unrolling
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int bb = 0; bb < 4; ++bb) fma4(acc[p][q], t.k[a][bb], v[2 * q + bb]);
            }
        }
        float* op = out + (size_t)b * OH * OW * C + 4 * c4;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = 0; q < 2; ++q)
                *reinterpret_cast<float4*>(op + ((size_t)(oy0 + p) * OW + ox0 + q) * C) = acc[p][q];
    }
}

int make_geom(int in_h, int in_w, int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1,
              int pad_y0, int pad_y1, const float* kernel_host, FirGeom* g, Taps* t) {
    if (in_h <= 0 || in_w <= 0 || kh <= 0 || kw <= 0 || kh * kw > 64 || !kernel_host) return EVC_EINVAL;
    if (up_x <= 0 || up_y <= 0 || down_x <= 0 || down_y <= 0) return EVC_EINVAL;
    if (pad_x0 < 0 || pad_x1 < 0 || pad_y0 < 0 || pad_y1 < 0) return EVC_EUNSUPPORTED;  // cropping pads unused here
    const int oh = (in_h * up_y + pad_y0 + pad_y1 - kh) / down_y + 1;
    const int ow = (in_w * up_x + pad_x0 + pad_x1 - kw) / down_x + 1;
    if (oh <= 0 || ow <= 0) return EVC_EINVAL;
    *g = FirGeom{in_h, in_w, oh, ow, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_y0};
    for (int a = 0; a < kh; ++a)
        for (int b = 0; b < kw; ++b) t->k[a * kw + b] = kernel_host[(kh - 1 - a) * kw + (kw - 1 - b)];
    return EVC_OK;
}

}  // namespace

extern "C" int evc_upfirdn2d_f32(const float* input, float* out, const float* kernel_host, int major, int in_h,
                                 int in_w, int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0,
                                 int pad_x1, int pad_y0, int pad_y1, void* stream) {
    if (!input || !out || major <= 0) return EVC_EINVAL;
    FirGeom g; Taps t;
    int rc = make_geom(in_h, in_w, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1, kernel_host,
                       &g, &t);
    if (rc != EVC_OK) return rc;
    const size_t total = (size_t)major * g.out_h * g.out_w;
    int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(upfirdn2d_nchw_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, input, out, t, g,
                       total);
    return hipGetLastError() == hipSuccess ? EVC_OK : EVC_ELAUNCH;
}

extern "C" int evc_upfirdn2d_nhwc_f32(const float* x, float* out, const float* kernel_host, int B, int H, int W,
                                      int C, int kh, int kw, int up, int down, int pad0, int pad1,
                                      const float* coef_a, const float* coef_s, int act, void* stream) {
    if (!x || !out || B <= 0 || C <= 0 || (C & 3) || ((coef_a == nullptr) != (coef_s == nullptr)))
        return EVC_EINVAL;
    FirGeom g; Taps t;
    int rc = make_geom(H, W, kh, kw, up, up, down, down, pad0, pad1, pad0, pad1, kernel_host, &g, &t);
    if (rc != EVC_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    // fast paths: the network's two resampling configurations
    const bool k44 = kh == 4 && kw == 4 && (act == EVC_ACT_NONE || act == EVC_ACT_SILU);
    if (k44 && up == 2 && down == 1 && pad0 == 2 && pad1 == 1 && !(H & 1) && !(W & 1)) {
        Taps16 t16;
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) t16.k[a][b] = t.k[a * 4 + b];
        const size_t total = (size_t)B * (H / 2) * (W / 2) * (C >> 2);
        const int grid = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
        if (act == EVC_ACT_SILU) hipLaunchKernelGGL((fir_up2_nhwc_kernel<EVC_ACT_SILU>), dim3(grid), dim3(256), 0, st, x, out, t16, B, H, W, C, coef_a, coef_s);
        else hipLaunchKernelGGL((fir_up2_nhwc_kernel<EVC_ACT_NONE>), dim3(grid), dim3(256), 0, st, x, out, t16, B, H, W, C, coef_a, coef_s);
        return hipGetLastError() == hipSuccess ? EVC_OK : EVC_ELAUNCH;
    }
    if (k44 && up == 1 && down == 2 && pad0 == 1 && pad1 == 1 && !(H & 3) && !(W & 3)) {
        Taps16 t16;
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) t16.k[a][b] = t.k[a * 4 + b];
        const size_t total = (size_t)B * (H / 4) * (W / 4) * (C >> 2);
        const int grid = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
        if (act == EVC_ACT_SILU) hipLaunchKernelGGL((fir_down2_nhwc_kernel<EVC_ACT_SILU>), dim3(grid), dim3(256), 0, st, x, out, t16, B, H, W, C, coef_a, coef_s);
        else hipLaunchKernelGGL((fir_down2_nhwc_kernel<EVC_ACT_NONE>), dim3(grid), dim3(256), 0, st, x, out, t16, B, H, W, C, coef_a, coef_s);
        return hipGetLastError() == hipSuccess ? EVC_OK : EVC_ELAUNCH;
    }
    const size_t total4 = (size_t)B * g.out_h * g.out_w * (C >> 2);
    int grid = (int)((total4 + 255) / 256 > 4096 ? 4096 : (total4 + 255) / 256);
    hipLaunchKernelGGL(upfirdn2d_nhwc_kernel, dim3(grid), dim3(256), 0, st, x, out, t, g, C,
                       coef_a, coef_s, act, total4);
    return hipGetLastError() == hipSuccess ? EVC_OK : EVC_ELAUNCH;
}
