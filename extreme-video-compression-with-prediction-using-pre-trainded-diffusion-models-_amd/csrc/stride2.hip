// stride2.hip -- 5x5 stride-2 transposed / strided convolutions of the ELIC codec as POLYPHASE 3x3 convolutions.
//
// compressai's deconv() = ConvTranspose2d(k 5, stride 2, padding 2, output_padding 1) and conv() = Conv2d(k 5, stride 2,
// padding 2) (reference Network.py:88-138 via compressai.models.utils).  Round 1 ran them as
// zero-insertion + 5x5 "same" convolution / 5x5 "same" convolution + decimation: 100 MACs per low-resolution pixel and
// channel pair where 25 are non-trivial.  Polyphase form, exact same sums:
//
//   transposed:  out[2i+py][2j+px] = sum_{a,b in -1..1} Wp[py][px][a][b] . x[i+a][j+b]
//                -> ONE 3x3 stride-1 convolution on the LOW-resolution input with 4*Co output channels (phase-major;
//                   the 2-tap phases carry a zero tap), then depth-to-space: 36 MACs
//   strided:     out[i][j] = sum_{a,b in -1..1} Wq[a][b][qy][qx] . x[2(i+a)+qy][2(j+b)+qx]
//                -> space-to-depth of the input (4*Ci channels), then ONE 3x3 stride-1 convolution: 36 MACs
//
// The 3x3 convolutions run through evc_conv2d_nhwc_f32 (row-reuse kernels); this file holds the two layout kernels, the
// weight re-arrangement and the two entry points SURVEY.md 8b lists (evc_deconv5x5s2_f32 / evc_conv5x5s2_f32).
#include <hip/hip_runtime.h>
#include "../../include/evc_hip.h"

namespace {

inline int grid_for(size_t n) {
    size_t g = (n + 255) / 256;
    return (int)(g > 4096 ? 4096 : (g ? g : 1));
}

// in [B][H][W][ld_in] with channel c4 = phase * Cp + c (phase = 2*py + px)  ->  out [B][2H][2W][C]
__global__ void depth_to_space2_kernel(const float* __restrict__ in, int ld_in, int Cp, float* __restrict__ out, int C,
                                       int H, int W, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        size_t t = i / C;
        const int X = (int)(t % (2 * W)); t /= 2 * W;
        const int Y = (int)(t % (2 * H));
        const int b = (int)(t / (2 * H));
        const int ph = 2 * (Y & 1) + (X & 1);
        out[i] = in[(((size_t)b * H + (Y >> 1)) * W + (X >> 1)) * ld_in + ph * Cp + c];
    }
}

// in [B][2H][2W][C]  ->  out [B][H][W][ld_out] with channel q * Cq + c (q = 2*qy + qx), channels c >= C of a phase zero
__global__ void space_to_depth2_kernel(const float* __restrict__ in, int ld_in, int C, float* __restrict__ out, int ld_out,
                                       int Cq, int H, int W, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cc = (int)(i % ld_out);
        size_t t = i / ld_out;
        const int x = (int)(t % W); t /= W;
        const int y = (int)(t % H);
        const int b = (int)(t / H);
        const int q = cc / Cq, c = cc - q * Cq;
        float v = 0.f;
        if (q < 4 && c < C) v = in[(((size_t)b * 2 * H + 2 * y + (q >> 1)) * 2 * W + 2 * x + (q & 1)) * ld_in + c];
        out[i] = v;
    }
}

// ConvTranspose2d weight wt [Ci][Co][5][5]  ->  3x3 correlation weight wp [4*Cp][CiPad][3][3]  (Cp >= Co, CiPad >= Ci):
// out[Y] = sum_ky wt[ky] x[(Y + 2 - ky) / 2] over ky == Y (mod 2); tap a in {0,1,2} reads x[i + a - 1]
//   py = 0: ky = 2 - 2(a-1) = 4, 2, 0        py = 1: a = 0 -> none, a = 1 -> ky = 3, a = 2 -> ky = 1
__global__ void deconv_phase_weights_kernel(const float* __restrict__ wt, float* __restrict__ wp, int Ci, int Co, int Cp,
                                            int CiPad, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i % 3);
        size_t t = i / 3;
        const int a = (int)(t % 3); t /= 3;
        const int ci = (int)(t % CiPad); t /= CiPad;
        const int oc = (int)t;                       // ph * Cp + co
        const int ph = oc / Cp, co = oc - ph * Cp;
        const int py = ph >> 1, px = ph & 1;
        const int ky = py == 0 ? 4 - 2 * a : (a == 0 ? -1 : 5 - 2 * a);
        const int kx = px == 0 ? 4 - 2 * b : (b == 0 ? -1 : 5 - 2 * b);
        float v = 0.f;
        if (ci < Ci && co < Co && ky >= 0 && kx >= 0) v = wt[(((size_t)ci * Co + co) * 5 + ky) * 5 + kx];
        wp[i] = v;
    }
}

// Conv2d weight w [Co][Ci][5][5]  ->  wq [Co][4*Cq][3][3] over the space-to-depth input (channel q * Cq + ci):
// out[i] = sum_u w[u] x[2i + u - 2];  2i + u - 2 = 2(i + a - 1) + qy  =>  u = 2a + qy  (u <= 4)
__global__ void conv_s2_phase_weights_kernel(const float* __restrict__ w, float* __restrict__ wq, int Co, int Ci, int Cq,
                                             size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i % 3);
        size_t t = i / 3;
        const int a = (int)(t % 3); t /= 3;
        const int cc = (int)(t % (4 * Cq)); t /= 4 * Cq;
        const int co = (int)t;
        const int q = cc / Cq, ci = cc - q * Cq;
        const int u = 2 * a + (q >> 1), v5 = 2 * b + (q & 1);
        float v = 0.f;
        if (ci < Ci && u < 5 && v5 < 5) v = w[(((size_t)co * Ci + ci) * 5 + u) * 5 + v5];
        wq[i] = v;
    }
}

}  // namespace

extern "C" int evc_depth_to_space2_f32(const float* in, int ld_in, int Cp, float* out, int C, int B, int H, int W,
                                       void* stream) {
    if (!in || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || Cp < C || ld_in < 4 * Cp) return EVC_EINVAL;
    const size_t total = (size_t)B * 2 * H * 2 * W * C;
    hipLaunchKernelGGL(depth_to_space2_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, ld_in, Cp, out,
                       C, H, W, total);
    return hipGetLastError() == hipSuccess ? EVC_OK : EVC_ELAUNCH;
}

extern "C" int evc_space_to_depth2_f32(const float* in, int ld_in, int C, float* out, int ld_out, int Cq, int B, int H, int W,
                                       void* stream) {
    if (!in || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || ld_in < C || Cq < C || ld_out < 4 * Cq) return EVC_EINVAL;
    const size_t total = (size_t)B * H * W * ld_out;
    hipLaunchKernelGGL(space_to_depth2_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, ld_in, C, out,
                       ld_out, Cq, H, W, total);
    return hipGetLastError() == hipSuccess ? EVC_OK : EVC_ELAUNCH;
}

extern "C" int evc_deconv5x5s2_phase_weights_f32(const float* wt, float* wp, int Ci, int Co, int Cp, int CiPad, void* stream) {
    if (!wt || !wp || Ci <= 0 || Co <= 0 || Cp < Co || CiPad < Ci) return EVC_EINVAL;
    const size_t total = (size_t)4 * Cp * CiPad * 9;
    hipLaunchKernelGGL(deconv_phase_weights_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, wt, wp, Ci, Co,
                       Cp, CiPad, total);
    return hipGetLastError() == hipSuccess ? EVC_OK : EVC_ELAUNCH;
}

extern "C" int evc_conv5x5s2_phase_weights_f32(const float* w, float* wq, int Co, int Ci, int Cq, void* stream) {
    if (!w || !wq || Ci <= 0 || Co <= 0 || Cq < Ci) return EVC_EINVAL;
    const size_t total = (size_t)Co * 4 * Cq * 9;
    hipLaunchKernelGGL(conv_s2_phase_weights_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, w, wq, Co, Ci,
                       Cq, total);
    return hipGetLastError() == hipSuccess ? EVC_OK : EVC_ELAUNCH;
}

extern "C" long long evc_deconv5x5s2_workspace_bytes(int B, int H, int W, int Cp) {
    if (B <= 0 || H <= 0 || W <= 0 || Cp <= 0) return EVC_EINVAL;
    return (long long)B * H * W * 4 * Cp * (long long)sizeof(float);
}

// x [B][H][W][Ci] (Ci % 16 == 0)  ->  out [B][2H][2W][Co].  w_packed: the (4*Cp, Ci, 3, 3) phase weights
// (evc_deconv5x5s2_phase_weights_f32) packed by evc_conv_pack_weights, Cp = Co rounded up to a multiple of 16;
// bias4: [4*Cp] = the layer's bias repeated per phase (zero in the padding), applied with act_out in the convolution's
// epilogue, so that the shuffle is a pure permutation; ws: evc_deconv5x5s2_workspace_bytes().
extern "C" int evc_deconv5x5s2_f32(const float* x, const void* w_packed, int arith, const float* bias4, float* out, float* ws,
                                   int B, int H, int W, int Ci, int Co, int act_out, void* stream) {
    if (!x || !w_packed || !out || !ws || Ci <= 0 || Ci % 16 != 0 || Co <= 0) return EVC_EINVAL;
    const int Cp = (Co + 15) / 16 * 16;
    evc_conv_args a = {};
    a.src0 = x; a.C0 = Ci; a.w_packed = reinterpret_cast<const float*>(w_packed); a.bias = bias4;
    a.out_scale = 1.0f; a.act_out = act_out; a.out = ws; a.ld_out = 4 * Cp;
    a.B = B; a.H = H; a.W = W; a.Co = 4 * Cp; a.KH = 3; a.KW = 3;
    a.splits = 1;                 // one fixed-order sum per output: entropy parameters must not depend on the launch
    a.arith = arith;
    const int rc = evc_conv2d_nhwc_f32(&a, nullptr, stream);
    if (rc != EVC_OK) return rc;
    return evc_depth_to_space2_f32(ws, 4 * Cp, Cp, out, Co, B, H, W, stream);
}

extern "C" long long evc_conv5x5s2_workspace_bytes(int B, int Ho, int Wo, int Ci) {
    if (B <= 0 || Ho <= 0 || Wo <= 0 || Ci <= 0) return EVC_EINVAL;
    const int Cq = (Ci + 3) / 4 * 4;
    return (long long)B * Ho * Wo * 4 * Cq * (long long)sizeof(float);
}

// x [B][2*Ho][2*Wo][ld_in] (first Ci channels)  ->  out [B][Ho][Wo][Co] = Conv2d(k 5, stride 2, padding 2).  w_packed: the (Co, 4*Cq, 3, 3) weights
// (evc_conv5x5s2_phase_weights_f32; Cq = Ci rounded up to a multiple of 4) packed by evc_conv_pack_weights.
extern "C" int evc_conv5x5s2_f32(const float* x, int ld_in, const void* w_packed, int arith, const float* bias, float* out,
                                 float* ws, int B, int Ho, int Wo, int Ci, int Co, int act_out, void* stream) {
    if (!x || !w_packed || !out || !ws || Ci <= 0 || Co <= 0 || ld_in < Ci) return EVC_EINVAL;
    const int Cq = (Ci + 3) / 4 * 4;
    int rc = evc_space_to_depth2_f32(x, ld_in, Ci, ws, 4 * Cq, Cq, B, Ho, Wo, stream);
    if (rc != EVC_OK) return rc;
    evc_conv_args a = {};
    a.src0 = ws; a.C0 = 4 * Cq; a.w_packed = reinterpret_cast<const float*>(w_packed); a.bias = bias;
    a.out_scale = 1.0f; a.act_out = act_out; a.out = out; a.ld_out = Co;
    a.B = B; a.H = Ho; a.W = Wo; a.Co = Co; a.KH = 3; a.KW = 3;
    a.splits = 1;
    a.arith = arith;
    return evc_conv2d_nhwc_f32(&a, nullptr, stream);
}
