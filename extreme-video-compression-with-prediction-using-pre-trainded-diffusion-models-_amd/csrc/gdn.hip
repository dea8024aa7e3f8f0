// gdn.hip -- Generalized Divisive Normalization (GDN / IGDN / GDN1) on NHWC tensors.
//
//   GDN   y[c] = x[c] * rsqrt(beta[c] + sum_j gamma[c][j] * x[j]^2)         IGDN: * sqrt(...)
//   GDN1  y[c] = x[c] / (beta[c] + sum_j gamma[c][j] * |x[j]|)              inverse: * (...)
//
// Replaces GDN.forward / GDN1.forward (reference ELICUtilis/layers/gdn.py:62-77, 95-106): a 1x1 convolution of the
// squared (absolute) input with the re-parametrised gamma as weights and beta as bias, then an elementwise rescale.
// TestModel.g_s / g_a contain no GDN (only ResidualBlockWithStride / Upsample use it, and Network.py
// imports neither), so this op is not on the decode hot path; it is the "next" row 8f-4 beside the alternative models.
// Three stream-ordered launches: square / abs pass (HBM-bound), the channel mixing on the matrix cores through
// evc_conv2d_nhwc_f32 (gamma packed like any 1x1 convolution weight; C % 16 == 0), rescale pass (HBM-bound).
#include <hip/hip_runtime.h>
#include "../../include/evc_hip.h"

namespace {

__global__ void gdn_pre_kernel(const float4* __restrict__ x, float4* __restrict__ y, size_t n4, int simplified) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = x[i];
        if (simplified) { v.x = fabsf(v.x); v.y = fabsf(v.y); v.z = fabsf(v.z); v.w = fabsf(v.w); }
        else { v.x *= v.x; v.y *= v.y; v.z *= v.z; v.w *= v.w; }
        y[i] = v;
    }
}

__device__ __forceinline__ float gdn_scale(float norm, int inverse, int simplified) {
    if (simplified) return inverse ? norm : 1.0f / norm;
    return inverse ? sqrtf(norm) : 1.0f / sqrtf(norm);
}

__global__ void gdn_post_kernel(const float4* __restrict__ x, const float4* __restrict__ norm, float4* __restrict__ out,
                                size_t n4, int inverse, int simplified) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = x[i], m = norm[i];
        out[i] = make_float4(v.x * gdn_scale(m.x, inverse, simplified), v.y * gdn_scale(m.y, inverse, simplified),
                             v.z * gdn_scale(m.z, inverse, simplified), v.w * gdn_scale(m.w, inverse, simplified));
    }
}

}  // namespace

extern "C" long long evc_gdn_workspace_bytes(int B, int H, int W, int C) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 16 != 0) return EVC_EINVAL;
    return 2LL * B * H * W * C * (long long)sizeof(float);      // x^2 (|x|) and the mixed norm
}

extern "C" int evc_gdn_f32(const float* x, const void* gamma_packed, int arith, const float* beta, float* out, float* ws,
                           int B, int H, int W, int C, int inverse, int simplified, void* stream) {
    if (!x || !gamma_packed || !beta || !out || !ws || evc_gdn_workspace_bytes(B, H, W, C) < 0) return EVC_EINVAL;
    if (arith == EVC_ARITH_F16X3) return EVC_EUNSUPPORTED;        // x^2 is unbounded: keep the exact bf16 split / f32
    const size_t n = (size_t)B * H * W * C, n4 = n / 4;
    float* sq = ws;
    float* norm = ws + n;
    const int grid = (int)((n4 + 255) / 256 > 4096 ? 4096 : (n4 + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gdn_pre_kernel, dim3(grid), dim3(256), 0, st, reinterpret_cast<const float4*>(x),
                       reinterpret_cast<float4*>(sq), n4, simplified);
    if (hipGetLastError() != hipSuccess) return EVC_ELAUNCH;
    evc_conv_args a = {};
    a.src0 = sq; a.C0 = C; a.w_packed = reinterpret_cast<const float*>(gamma_packed); a.bias = beta;
    a.out_scale = 1.0f; a.out = norm; a.ld_out = C; a.B = B; a.H = H; a.W = W; a.Co = C; a.KH = 1; a.KW = 1;
    a.splits = 1; a.arith = arith;
    const int rc = evc_conv2d_nhwc_f32(&a, nullptr, stream);
    if (rc != EVC_OK) return rc;
    hipLaunchKernelGGL(gdn_post_kernel, dim3(grid), dim3(256), 0, st, reinterpret_cast<const float4*>(x),
                       reinterpret_cast<const float4*>(norm), reinterpret_cast<float4*>(out), n4, inverse, simplified);
    return hipGetLastError() == hipSuccess ? EVC_OK : EVC_ELAUNCH;
}
