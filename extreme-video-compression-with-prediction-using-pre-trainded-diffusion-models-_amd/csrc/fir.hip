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
    const size_t total4 = (size_t)B * g.out_h * g.out_w * (C >> 2);
    int grid = (int)((total4 + 255) / 256 > 4096 ? 4096 : (total4 + 255) / 256);
    hipLaunchKernelGGL(upfirdn2d_nhwc_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, out, t, g, C,
                       coef_a, coef_s, act, total4);
    return hipGetLastError() == hipSuccess ? EVC_OK : EVC_ELAUNCH;
}
