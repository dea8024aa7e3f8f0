// conv_rr_k32_r03.hip -- EXPERIMENT (not part of the library): the row-reuse 3x3 convolution of csrc/conv_igemm.hip on the
// v_mfma_f32_16x16x32_f16 shape.  At the package power cap that shape sustains 1.19 x the FLOP/s of the 32x32x16 shape the
// product kernel uses (tools/mfma_shape_probe.hip, profiles/r03_mfma_shape_probe.log); this file measures how much of that
// survives in the real loop (LDS fragments, LDS-DMA weights, GroupNorm+SiLU staging) and checks the result against the product.
//
// Same arithmetic (f16x3: operands scaled by powers of two, 2-way fp16 split, 3 MFMAs per product, fp32 accumulate), same
// structure (macro-step = (channel chunk, kernel row): the two image rows of the tile staged once, three horizontal taps =
// three K-steps reading that image at shifted rows; weights per tap by LDS-DMA, double-buffered), but
//   * K-step = 32 channels (the instruction's K): LDS rows of 64 B, four 16-byte slots per row, slot s of row r stored at
//     s ^ ((r >> 2) & 3) (conflict-free for the 16-lane groups of ds_read_b128);
//   * tile 256 pixels x 192 channels, 8 waves (4 x 2), wave tile 64 x 96 = 4 x 6 accumulator tiles of 4 registers;
//     LDS 2 x 2 x (260 + 192) x 64 B = 115.7 KB, one workgroup per CU (two 4-wave workgroups would need 164.9 KB);
//   * weights packed [tap][Ci/32][plane][Co][64 B pre-swizzled] on the host here (the library packs 16-channel chunks).
// Restricted to what the measurement needs: 128-wide images, one source tensor, Ci % 32 == 0, Co = 192, unsplit grids,
// input transform plain or GroupNorm-affine + SiLU, bias epilogue.  Simple schedule (vmcnt(0) + one barrier per K-step).
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/experiments/conv_rr_k32_r03.hip -o tools/conv_bench_k32
// Run:   ./tools/conv_bench_k32 [iters]     (B = 8: 512 tiles = 2 rounds; Ci = 192 and 384; random operands)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../extreme-video-compression-with-prediction-using-pre-trainded-diffusion-models-_amd/csrc/conv_igemm.hip"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));

struct K32Args {
    const float* x; const char* w; const float* coef_a; const float* coef_s; const float* bias; float* out;
    int B, H, W, Ci, Co; float ascale;       // ascale = 1 / (S_a S_w)
};

template <bool COEF>
__global__ __launch_bounds__(512, 1) void conv_rr_k32_kernel(K32Args p) {
    constexpr int BM = 256, BN = 192, RB = 64, NT = 512;
    constexpr int WPL = BN * RB;                       // bytes per weight plane of a K-step
    extern __shared__ __attribute__((aligned(16))) char smem_k[];
    const int W = p.W;
    const int SR = (BM / W) * (W + 2);                 // staged rows: the tile's image rows, each with a zero halo pixel per side
    const int APL = SR * RB;                           // bytes per activation plane
    char* const As = smem_k;                           // [2 buffers][2 planes][SR][64 B]
    char* const Ws = smem_k + 4 * APL;                 // [2 buffers][2 planes][192][64 B]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r16 = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * BM;
    const int HW = p.H * W;
    const int nchunk = p.Ci / 32, nmac = nchunk * 3, nstep = nmac * 3;

    // ---- staging thread: pixel `row` of the tile, channels 16 kh .. 16 kh + 15 of the chunk ----
    const int row = tid >> 1, kh = tid & 1;
    const int m = m0 + row;
    const int b = m / HW, rem = m - b * HW, y = rem / W;
    const int srow_p = row + 2 * (row / W) + 1;
    const int sw_p = (srow_p >> 2) & 3;
    const int a_st0 = srow_p * RB + 16 * ((2 * kh) ^ sw_p), a_st1 = srow_p * RB + 16 * ((2 * kh + 1) ^ sw_p);
    const float* xrow = p.x + (size_t)m * p.Ci + 16 * kh;       // this pixel, channel 16 kh of chunk 0

    // ---- fragment read offsets ----
    int ard[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int px = wm * 64 + i * 16 + r16;
        const int base = px + 2 * (px / W);
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) {
            const int sr = base + tx;
            ard[i][tx] = sr * RB + 16 * (g ^ ((sr >> 2) & 3));
        }
    }
    int brd[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int co = wn * 96 + j * 16 + r16;
        brd[j] = co * RB + 16 * (g ^ ((co >> 2) & 3));
    }

    // zero both activation images once: the halo pixels stay zero for the whole kernel
    for (int o = tid * 16; o < 4 * APL; o += NT * 16) *reinterpret_cast<float4*>(As + o) = make_float4(0.f, 0.f, 0.f, 0.f);

    float4 areg[4], ca[4], cs[4];
    bool aok = false;
    auto load_coefs = [&](int chunk) {
        if (COEF) {
            const size_t co = (size_t)b * p.Ci + chunk * 32 + 16 * kh;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                ca[q] = *reinterpret_cast<const float4*>(p.coef_a + co + 4 * q);
                cs[q] = *reinterpret_cast<const float4*>(p.coef_s + co + 4 * q);
            }
        }
    };
    auto load_a = [&](int chunk, int ty) {
        const int yy = y + ty - 1;
        aok = yy >= 0 && yy < p.H;
        const float* src = aok ? xrow + (ptrdiff_t)(ty - 1) * W * p.Ci + chunk * 32 : p.x + 16 * kh;
#pragma unroll
        for (int q = 0; q < 4; ++q) areg[q] = *reinterpret_cast<const float4*>(src + 4 * q);
    };
    auto store_a = [&](int ab) {
        f16x8 h0, l0, h1, l1;
        constexpr int MODE = COEF ? MODE_AFFINE_SILU : MODE_PLAIN;
        split2_f16(transform<MODE>(areg[0], ca[0], cs[0], aok), transform<MODE>(areg[1], ca[1], cs[1], aok), F16_ACT_SCALE, h0, l0);
        split2_f16(transform<MODE>(areg[2], ca[2], cs[2], aok), transform<MODE>(areg[3], ca[3], cs[3], aok), F16_ACT_SCALE, h1, l1);
        char* A = As + ab * 2 * APL;
        *reinterpret_cast<f16x8*>(A + a_st0) = h0; *reinterpret_cast<f16x8*>(A + APL + a_st0) = l0;
        *reinterpret_cast<f16x8*>(A + a_st1) = h1; *reinterpret_cast<f16x8*>(A + APL + a_st1) = l1;
    };
    auto dma_w = [&](int step, int wb) {               // K-step `step` = (macro-step, tap): slab ((ty*3 + tx) * nchunk + chunk)
        const int mac = step / 3, tx = step - mac * 3, chunk = mac / 3, ty = mac - chunk * 3;
        const char* src = p.w + (size_t)((ty * 3 + tx) * nchunk + chunk) * (2 * WPL);
        char* dst = Ws + wb * 2 * WPL;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int piece = wave * 3 + j;            // 24 pieces of 1 KiB, 3 per wave
            __builtin_amdgcn_global_load_lds((glb_void*)(src + piece * 1024 + lane * 16), (lds_void*)(dst + piece * 1024), 16, 0, 0);
        }
    };

    f32x4v acc[4][6];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    load_coefs(0);
    load_a(0, 0);
    dma_w(0, 0);
    __syncthreads();                                    // images zeroed
    store_a(0);

    for (int mac = 0; mac < nmac; ++mac) {
        const int ab = mac & 1;
        const int nchk = (mac + 1) / 3, nty = (mac + 1) - nchk * 3;
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) {
            const int step = mac * 3 + tx, wb = step & 1;
            __builtin_amdgcn_s_waitcnt(0);              // this step's weight pieces landed (and everything older)
            __syncthreads();
            if (step + 1 < nstep) dma_w(step + 1, wb ^ 1);
            if (tx == 0 && mac + 1 < nmac) {
                if (nty == 0) load_coefs(nchk);
                load_a(nchk, nty);
            }
            const char* Ab = As + ab * 2 * APL;
            const char* Wb = Ws + wb * 2 * WPL;
            f16x8 a[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i][0] = *reinterpret_cast<const f16x8*>(Ab + ard[i][tx]);
                a[i][1] = *reinterpret_cast<const f16x8*>(Ab + APL + ard[i][tx]);
            }
#pragma unroll
            for (int jh = 0; jh < 2; ++jh) {            // three channel tiles at a time: 24 fragment registers instead of 48
                f16x8 bq[3][2];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    bq[j][0] = *reinterpret_cast<const f16x8*>(Wb + brd[3 * jh + j]);
                    bq[j][1] = *reinterpret_cast<const f16x8*>(Wb + WPL + brd[3 * jh + j]);
                }
                // terms as in Split<2>: (a_lo, w_hi), (a_hi, w_lo), (a_hi, w_hi)
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 3; ++j)
                            acc[i][3 * jh + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i][t == 0 ? 1 : 0], bq[j][t == 1 ? 1 : 0],
                                                                                       acc[i][3 * jh + j], 0, 0, 0);
            }
            if (tx == 2 && mac + 1 < nmac) store_a(ab ^ 1);
        }
    }

    // epilogue: C layout of the 16x16 MFMA: column = lane & 15 (channel), rows 4 (lane >> 4) + r (pixels)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int co = wn * 96 + j * 16 + r16;
        const float bias = p.bias ? p.bias[co] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float* o = p.out + (size_t)(m0 + wm * 64 + i * 16 + 4 * g) * p.Co + co;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[(size_t)r * p.Co] = acc[i][j][r] * p.ascale + bias;
        }
    }
}

// host pack: raw (Co, Ci, 3, 3) fp32 -> [tap][Ci/32][plane][Co][4 slots pre-swizzled][8 halves]; returns S_w
float pack_k32(const std::vector<float>& w, int Co, int Ci, std::vector<_Float16>& out) {
    float mx = 0.f;
    for (float v : w) mx = std::fmax(mx, std::fabs(v));
    const float Sw = mx > 0.f ? std::ldexp(1.0f, 14 - std::ilogb(mx)) : 1.0f;     // max |w| S_w in [2^14, 2^15)
    const int nchunk = Ci / 32;
    out.assign((size_t)9 * nchunk * 2 * Co * 32, (_Float16)0.f);
    for (int tap = 0; tap < 9; ++tap)
        for (int c = 0; c < nchunk; ++c)
            for (int co = 0; co < Co; ++co)
                for (int k = 0; k < 32; ++k) {
                    const float v = w[((size_t)co * Ci + c * 32 + k) * 9 + tap] * Sw;
                    const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
                    const int slot = (k >> 3) ^ ((co >> 2) & 3);
                    const size_t base = ((size_t)(tap * nchunk + c) * 2) * Co * 32;
                    out[base + (size_t)co * 32 + slot * 8 + (k & 7)] = hi;
                    out[base + (size_t)Co * 32 + (size_t)co * 32 + slot * 8 + (k & 7)] = lo;
                }
    return Sw;
}

}  // namespace

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    const int B = 8, R = 128, Co = 192;
    g_tail_split = 1; g_wide_tiles = 1;
    for (int Ci : {192, 384})
        for (int mode = 0; mode < 2; ++mode) {
            const size_t nx = (size_t)B * R * R * Ci, no = (size_t)B * R * R * Co, nraw = (size_t)Co * Ci * 9;
            std::vector<float> hx(nx), hw(nraw);
            for (size_t i = 0; i < nx; ++i) hx[i] = (float)((int)((i * 2654435761u) >> 8 & 0xffff) - 32768) / 32768.0f;
            for (size_t i = 0; i < nraw; ++i) hw[i] = 0.02f * (float)((int)(((i + 977) * 2246822519u) >> 8 & 0xffff) - 32768) / 32768.0f;
            float *x, *wraw, *o_ref, *o_new, *ca, *cs; void* wp; char* w32;
            CK(hipMalloc(&x, nx * 4)); CK(hipMalloc(&wraw, nraw * 4)); CK(hipMalloc(&o_ref, no * 4)); CK(hipMalloc(&o_new, no * 4));
            CK(hipMalloc(&ca, (size_t)B * Ci * 4)); CK(hipMalloc(&cs, (size_t)B * Ci * 4));
            CK(hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(wraw, hw.data(), nraw * 4, hipMemcpyHostToDevice));
            std::vector<float> one((size_t)B * Ci, 1.0f), shift((size_t)B * Ci, 0.1f);
            CK(hipMemcpy(ca, one.data(), one.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(cs, shift.data(), shift.size() * 4, hipMemcpyHostToDevice));
            // product kernel (reference result + timing)
            CK(hipMalloc(&wp, (size_t)evc_conv_packed_bytes(Co, Ci, 3, 3, EVC_ARITH_F16X3)));
            if (evc_conv_pack_weights(wraw, wp, Co, Ci, 3, 3, EVC_ARITH_F16X3, nullptr) != 0) { printf("pack failed\n"); return 1; }
            evc_conv_args a = {};
            a.src0 = x; a.C0 = Ci; a.w_packed = (const float*)wp; a.out = o_ref; a.ld_out = Co; a.out_scale = 1.f;
            a.B = B; a.H = R; a.W = R; a.Co = Co; a.KH = 3; a.KW = 3; a.arith = EVC_ARITH_F16X3;
            if (mode) { a.coef_a = ca; a.coef_s = cs; a.act_in = EVC_ACT_SILU; }
            long long wsb = evc_conv_workspace_bytes(&a);
            float* ws = nullptr; if (wsb > 0) CK(hipMalloc(&ws, wsb));
            // experiment kernel
            std::vector<_Float16> packed;
            const float Sw = pack_k32(hw, Co, Ci, packed);
            CK(hipMalloc(&w32, packed.size() * 2)); CK(hipMemcpy(w32, packed.data(), packed.size() * 2, hipMemcpyHostToDevice));
            K32Args k = {x, w32, mode ? ca : nullptr, mode ? cs : nullptr, nullptr, o_new, B, R, R, Ci, Co, 1.0f / (Sw * F16_ACT_SCALE)};
            const size_t lds = (size_t)4 * (256 / R) * (R + 2) * 64 + (size_t)4 * 192 * 64;
            const void* fn = mode ? reinterpret_cast<const void*>(&conv_rr_k32_kernel<true>) : reinterpret_cast<const void*>(&conv_rr_k32_kernel<false>);
            CK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const dim3 grid((unsigned)((size_t)B * R * R / 256));
            auto run_new = [&]() {
                if (mode) hipLaunchKernelGGL(conv_rr_k32_kernel<true>, grid, dim3(512), lds, 0, k);
                else hipLaunchKernelGGL(conv_rr_k32_kernel<false>, grid, dim3(512), lds, 0, k);
            };
            if (evc_conv2d_nhwc_f32(&a, ws, nullptr) != 0) { printf("product launch failed\n"); return 1; }
            run_new();
            CK(hipDeviceSynchronize());
            std::vector<float> r1(no), r2(no);
            CK(hipMemcpy(r1.data(), o_ref, no * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(r2.data(), o_new, no * 4, hipMemcpyDeviceToHost));
            double mx = 0, sc = 0; size_t bad = 0;
            for (size_t i = 0; i < no; ++i) { const double d = std::fabs((double)r1[i] - r2[i]); if (!(d <= 1e30)) ++bad; mx = std::fmax(mx, d); sc = std::fmax(sc, std::fabs((double)r1[i])); }
            double us[2];
            for (int which = 0; which < 2; ++which) {
                hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                CK(hipEventRecord(e0));
                for (int i = 0; i < iters; ++i) { if (which == 0) evc_conv2d_nhwc_f32(&a, ws, nullptr); else run_new(); }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); us[which] = ms * 1e3 / iters;
            }
            const double gf = 2.0 * B * R * R * (double)Ci * Co * 9 / 1e9;
            printf("B=%d 128x128 %d->192 k3 %s: product (32x32x16) %.1f us = %.1f TF/s | K=32 experiment (16x16x32) %.1f us = %.1f TF/s | x%.3f | max|diff|/max|out| %.2e%s\n",
                   B, Ci, mode ? "gn+silu" : "plain  ", us[0], gf / us[0] * 1e3 / 1e3 * 1.0, us[1], gf / us[1] * 1.0, us[0] / us[1], mx / sc, bad ? "  NON-FINITE" : "");
            fflush(stdout);
            hipFree(x); hipFree(wraw); hipFree(o_ref); hipFree(o_new); hipFree(ca); hipFree(cs); hipFree(wp); hipFree(w32); if (ws) hipFree(ws);
        }
    return 0;
}
