// conv_igemm.hip -- stride-1 "same" convolution as an implicit GEMM on the gfx950 f32 matrix cores.
//
//   D[m][co] = sum_{tap, ci} act(src[m shifted by tap][ci] * a[b][ci] + s[b][ci]) * W[co][ci][tap]
//
// GEMM view: M = B*H*W output pixels (A operand, gathered from NHWC activations), N = Co (B operand,
// pre-packed weights), K = KH*KW*Ci walked in steps of KC = 16 channels of one tap.
// One workgroup = 4 waves (2 x 2) computes a 128 x (64*TN) tile with v_mfma_f32_32x32x2_f32
// (exact f32: bitwise a k-ordered fmaf chain); each wave owns 64 pixels x 32*TN channels.
// Per step the A tile [128][16] and W tile [64*TN][16] are prefetched global -> registers while the
// previous step's MFMAs run, transformed (GroupNorm affine + SiLU/ReLU, zero padding after the
// activation) and written to the other LDS buffer: one barrier per step.
// LDS rows are padded to 20 floats so the ds_read_b128 fragment reads (4 k-values per lane: lanes
// 0-31 take k 0..3, lanes 32-63 take k 4..7 of each 8-deep group) are bank-conflict free.
//
// Replaces nn.Conv2d 3x3/1x1 (reference models/better/layers.py:89-113), NIN (layers.py:535-544),
// nn.Linear (ncsnpp_more.py:89-95, layerspp.py:507) and the ELIC conv stacks (Network.py:106-166).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/evc_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BM = 128;      // pixels per workgroup tile
constexpr int KC = 16;       // channels per K step
constexpr int LDS_LD = 20;   // padded LDS row (floats)

struct ConvK {
    const float* src0; const float* src1; int C0; int C1; int ld0; int ld1;
    const float* coef_a; const float* coef_s; int act_in;
    const float* w; const float* bias; const float* res; int ld_res;
    float out_scale; int act_out;
    float* out; int ld_out;
    int B, H, W, Co, CoPad, KH, KW;
    int M, HW, nchunk, nsteps, steps_per_split, splits;
    float* ws;   // split-K slabs [splits][M][Co] when splits > 1
};

__device__ __forceinline__ float act_fn(float v, int act) {
    if (act == EVC_ACT_SILU) return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
    if (act == EVC_ACT_RELU) return fmaxf(v, 0.0f);
    return v;
}

template <int TN>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(ConvK p) {
    constexpr int BN = 64 * TN;
    constexpr int WLOADS = BN / 64;   // float4 weight loads per thread per step
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const As = smem;                        // [2][BM][LDS_LD]
    float* const Ws = smem + 2 * BM * LDS_LD;      // [2][BN][LDS_LD]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, half = lane >> 5;

    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int split = blockIdx.z;
    const int s_begin = split * p.steps_per_split;
    const int s_end = min(p.nsteps, s_begin + p.steps_per_split);

    // ---- per-thread gather bookkeeping: two A rows (r, r + 64), one 4-channel column k4 ----
    const int k4 = tid & 3;
    const int padH = p.KH >> 1, padW = p.KW >> 1;
    const int taps = p.KH * p.KW;
    int rb[2], ry[2], rx[2];
    bool rvalid[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int m = m0 + (tid >> 2) + 64 * i;
        rvalid[i] = m < p.M;
        int mm = rvalid[i] ? m : 0;
        int b = mm / p.HW;
        int rem = mm - b * p.HW;
        int y = rem / p.W;
        rb[i] = b; ry[i] = y; rx[i] = rem - y * p.W;
    }
    const int Ct = p.C0 + p.C1;
    const bool has_coef = p.coef_a != nullptr;

    float4 areg[2], ca[2], cs[2], wreg[WLOADS];
    bool aok[2];

    auto issue_loads = [&](int s) {
        const int chunk = s / taps;
        const int tap = s - chunk * taps;
        const int dy = tap / p.KW - padH;
        const int dx = tap - (tap / p.KW) * p.KW - padW;
        const int c = chunk * KC;
        const float* src; int cs_off, Csrc;
        if (c < p.C0) { src = p.src0; cs_off = c; Csrc = p.ld0; }
        else { src = p.src1; cs_off = c - p.C0; Csrc = p.ld1; }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int yy = ry[i] + dy, xx = rx[i] + dx;
            aok[i] = rvalid[i] && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
            if (aok[i]) {
                const size_t off = ((size_t)(rb[i] * p.H + yy) * p.W + xx) * Csrc + cs_off + 4 * k4;
                areg[i] = *reinterpret_cast<const float4*>(src + off);
                if (has_coef) {
                    const size_t co = (size_t)rb[i] * Ct + c + 4 * k4;
                    ca[i] = *reinterpret_cast<const float4*>(p.coef_a + co);
                    cs[i] = *reinterpret_cast<const float4*>(p.coef_s + co);
                }
            } else {
                areg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        const float* wt = p.w + ((size_t)(tap * p.nchunk + chunk) * p.CoPad + n0) * KC;
#pragma unroll
        for (int j = 0; j < WLOADS; ++j)
            wreg[j] = *reinterpret_cast<const float4*>(wt + (size_t)(tid + 256 * j) * 4);
    };

    auto store_tiles = [&](int buf) {
        float* A = As + buf * BM * LDS_LD;
        float* Wl = Ws + buf * BN * LDS_LD;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float4 v = areg[i];
            if (aok[i]) {
                if (has_coef) {
                    v.x = v.x * ca[i].x + cs[i].x; v.y = v.y * ca[i].y + cs[i].y;
                    v.z = v.z * ca[i].z + cs[i].z; v.w = v.w * ca[i].w + cs[i].w;
                }
                if (p.act_in != EVC_ACT_NONE) {
                    v.x = act_fn(v.x, p.act_in); v.y = act_fn(v.y, p.act_in);
                    v.z = act_fn(v.z, p.act_in); v.w = act_fn(v.w, p.act_in);
                }
            }
            *reinterpret_cast<float4*>(A + ((tid >> 2) + 64 * i) * LDS_LD + 4 * k4) = v;
        }
#pragma unroll
        for (int j = 0; j < WLOADS; ++j) {
            const int idx = tid + 256 * j;
            *reinterpret_cast<float4*>(Wl + (idx >> 2) * LDS_LD + 4 * (idx & 3)) = wreg[j];
        }
    };

    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (s_begin < s_end) {
        issue_loads(s_begin);
        store_tiles(0);
    }
    __syncthreads();

    for (int s = s_begin; s < s_end; ++s) {
        const int buf = (s - s_begin) & 1;
        const bool more = (s + 1) < s_end;
        if (more) issue_loads(s + 1);

        const float* Ab = As + buf * BM * LDS_LD + (wm * 64 + l31) * LDS_LD + 4 * half;
        const float* Wb = Ws + buf * BN * LDS_LD + (wn * 32 * TN + l31) * LDS_LD + 4 * half;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            float4 a[2], b[TN];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const float4*>(Ab + i * 32 * LDS_LD + kk * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const float4*>(Wb + j * 32 * LDS_LD + kk * 8);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (more) store_tiles(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: C/D map of the 32x32 MFMA: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
    const bool partial = p.splits > 1;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int co = n0 + wn * 32 * TN + j * 32 + l31;
        if (co >= p.Co) continue;
        const float bias = (!partial && p.bias) ? p.bias[co] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m >= p.M) continue;
                float v = acc[i][j][r];
                if (partial) {
                    p.ws[((size_t)split * p.M + m) * p.Co + co] = v;
                } else {
                    v += bias;
                    if (p.res) v += p.res[(size_t)m * p.ld_res + co];
                    v *= p.out_scale;
                    v = act_fn(v, p.act_out);
                    p.out[(size_t)m * p.ld_out + co] = v;
                }
            }
        }
    }
}

// out = act((sum_z ws[z] + bias + res) * scale): deterministic split-K combine.
__global__ void conv_splitk_reduce_kernel(const float* ws, int splits, int M, int Co, const float* bias,
                                          const float* res, int ld_res, float scale, int act, float* out,
                                          int ld_out) {
    const size_t total = (size_t)M * Co;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / Co), co = (int)(i - (size_t)m * Co);
        float v = 0.f;
        for (int z = 0; z < splits; ++z) v += ws[(size_t)z * total + i];
        if (bias) v += bias[co];
        if (res) v += res[(size_t)m * ld_res + co];
        v *= scale;
        out[(size_t)m * ld_out + co] = act_fn(v, act);
    }
}

// w [Co][Ci][KH][KW] -> packed [KH*KW][Ci/16][CoPad][16] (zero rows for co >= Co).
__global__ void conv_pack_weights_kernel(const float* w, float* packed, int Co, int CoPad, int Ci, int KH, int KW) {
    const int taps = KH * KW, nchunk = Ci / KC;
    const size_t total = (size_t)taps * nchunk * CoPad * KC;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % KC);
        size_t t = i / KC;
        const int co = (int)(t % CoPad); t /= CoPad;
        const int chunk = (int)(t % nchunk);
        const int tap = (int)(t / nchunk);
        float v = 0.f;
        if (co < Co) v = w[((size_t)co * Ci + chunk * KC + k) * taps + tap];
        packed[i] = v;
    }
}

int pick_tn(int CoPad) {
    if (CoPad % 192 == 0) return 3;
    if (CoPad % 128 == 0) return 2;
    return 1;
}

}  // namespace

extern "C" int evc_conv_co_pad(int Co) { return (Co + 63) / 64 * 64; }

extern "C" long long evc_conv_packed_floats(int Co, int Ci, int KH, int KW) {
    return (long long)KH * KW * (Ci / KC) * evc_conv_co_pad(Co) * KC;
}

extern "C" int evc_conv_pack_weights_f32(const float* w, float* packed, int Co, int Ci, int KH, int KW, void* stream) {
    if (!w || !packed || Co <= 0 || Ci <= 0 || Ci % KC != 0 || KH <= 0 || KW <= 0) return EVC_EINVAL;
    const long long total = evc_conv_packed_floats(Co, Ci, KH, KW);
    int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(conv_pack_weights_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, packed, Co,
                       evc_conv_co_pad(Co), Ci, KH, KW);
    return hipGetLastError() == hipSuccess ? EVC_OK : EVC_ELAUNCH;
}

static int conv_validate(const evc_conv_args* a) {
    if (!a || !a->src0 || !a->w_packed || !a->out) return EVC_EINVAL;
    if (a->C0 <= 0 || a->C0 % KC != 0 || a->C1 < 0 || a->C1 % KC != 0) return EVC_EINVAL;
    if (a->C1 > 0 && !a->src1) return EVC_EINVAL;
    if ((a->ld0 != 0 && (a->ld0 < a->C0 || (a->ld0 & 3))) || (a->ld1 != 0 && (a->ld1 < a->C1 || (a->ld1 & 3))))
        return EVC_EINVAL;
    if ((a->coef_a == nullptr) != (a->coef_s == nullptr)) return EVC_EINVAL;
    if (a->B <= 0 || a->H <= 0 || a->W <= 0 || a->Co <= 0) return EVC_EINVAL;
    if (a->KH <= 0 || a->KW <= 0 || !(a->KH & 1) || !(a->KW & 1)) return EVC_EINVAL;
    if (a->ld_out < a->Co || (a->res && a->ld_res < a->Co)) return EVC_EINVAL;
    if ((long long)a->B * a->H * a->W > 0x7fffffffLL) return EVC_EINVAL;
    return EVC_OK;
}

extern "C" int evc_conv_choose_splits(const evc_conv_args* a) {
    if (conv_validate(a) != EVC_OK) return EVC_EINVAL;
    if (a->splits > 0) return a->splits;
    const int M = a->B * a->H * a->W;
    const int CoPad = evc_conv_co_pad(a->Co);
    const int BN = 64 * pick_tn(CoPad);
    const long long tiles = (long long)((M + BM - 1) / BM) * (CoPad / BN);
    const int nsteps = a->KH * a->KW * ((a->C0 + a->C1) / KC);
    // Fill the chip (256 CUs x 2 resident workgroups) when the tile grid alone cannot, but keep at
    // least 8 K-steps per split so the prologue/epilogue stay amortised.
    int splits = 1;
    if (tiles < 384) {
        splits = (int)((512 + tiles - 1) / tiles);
        const int max_by_steps = nsteps / 8 > 0 ? nsteps / 8 : 1;
        if (splits > max_by_steps) splits = max_by_steps;
        if (splits > 32) splits = 32;
        if (splits < 1) splits = 1;
    }
    return splits;
}

extern "C" long long evc_conv_workspace_bytes(const evc_conv_args* a) {
    const int s = evc_conv_choose_splits(a);
    if (s < 0) return s;
    if (s == 1) return 0;
    return (long long)s * a->B * a->H * a->W * a->Co * (long long)sizeof(float);
}

extern "C" int evc_conv2d_nhwc_f32(const evc_conv_args* a, float* ws, void* stream) {
    int rc = conv_validate(a);
    if (rc != EVC_OK) return rc;
    ConvK k;
    k.src0 = a->src0; k.src1 = a->src1; k.C0 = a->C0; k.C1 = a->C1;
    k.ld0 = a->ld0 > 0 ? a->ld0 : a->C0; k.ld1 = a->ld1 > 0 ? a->ld1 : a->C1;
    k.coef_a = a->coef_a; k.coef_s = a->coef_s; k.act_in = a->act_in;
    k.w = a->w_packed; k.bias = a->bias; k.res = a->res; k.ld_res = a->ld_res;
    k.out_scale = a->out_scale; k.act_out = a->act_out; k.out = a->out; k.ld_out = a->ld_out;
    k.B = a->B; k.H = a->H; k.W = a->W; k.Co = a->Co; k.CoPad = evc_conv_co_pad(a->Co);
    k.KH = a->KH; k.KW = a->KW;
    k.M = a->B * a->H * a->W; k.HW = a->H * a->W;
    k.nchunk = (a->C0 + a->C1) / KC;
    k.nsteps = a->KH * a->KW * k.nchunk;
    k.splits = evc_conv_choose_splits(a);
    k.steps_per_split = (k.nsteps + k.splits - 1) / k.splits;
    k.splits = (k.nsteps + k.steps_per_split - 1) / k.steps_per_split;   // no empty splits
    k.ws = ws;
    if (k.splits > 1 && !ws) return EVC_EINVAL;

    const int tn = pick_tn(k.CoPad);
    const int BN = 64 * tn;
    dim3 grid((k.M + BM - 1) / BM, k.CoPad / BN, k.splits);
    const size_t lds = (size_t)2 * (BM + BN) * LDS_LD * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (tn == 3) hipLaunchKernelGGL(conv_igemm_kernel<3>, grid, dim3(256), lds, st, k);
    else if (tn == 2) hipLaunchKernelGGL(conv_igemm_kernel<2>, grid, dim3(256), lds, st, k);
    else hipLaunchKernelGGL(conv_igemm_kernel<1>, grid, dim3(256), lds, st, k);
    if (hipGetLastError() != hipSuccess) return EVC_ELAUNCH;
    if (k.splits > 1) {
        const size_t total = (size_t)k.M * k.Co;
        int g = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
        hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3(g), dim3(256), 0, st, ws, k.splits, k.M, k.Co, a->bias,
                           a->res, a->ld_res, a->out_scale, a->act_out, a->out, a->ld_out);
        if (hipGetLastError() != hipSuccess) return EVC_ELAUNCH;
    }
    return EVC_OK;
}
