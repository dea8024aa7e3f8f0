// norm.hip -- GroupNorm as (per-channel moments) -> (per-(b,c) affine coefficients).
//
// The normalisation itself is never a pass of its own: consumers (conv_igemm, upfirdn2d_nhwc) apply
// x*coef_a + coef_s (+ SiLU) while loading.  Moments are kept PER CHANNEL so that the same numbers serve
// any grouping, including groups that straddle the two tensors of a skip-connection concat
// (reference models/better/ncsnpp_more.py:362-364 feeds cat[h, skip] to get_act_norm).
// HBM-bound: chan_stats reads the tensor once (4 B/element), everything else is O(B*C).
//
// Reference semantics: nn.GroupNorm (biased variance) via models/better/layerspp.py:473-477 (eps 1e-5),
// :215 (eps 1e-6, affine), AdaGN get_act_norm.forward :518-549.
#include <hip/hip_runtime.h>
#include "../../include/evc_hip.h"

namespace {

// grid (nsplit, B), block 256.  Thread t owns channel quad c4 = t % C4 (C4 = C/4) and walks pixels
// p = t / C4, + R, ... of the block's pixel range (R = 256 / C4 rows in flight); for C4 > 256 a thread
// owns several quads.  Wave-coalesced float4 loads along C.
__global__ __launch_bounds__(256) void chan_stats_kernel(const float* __restrict__ x, float* __restrict__ partial,
                                                         int HW, int C, int nsplit) {
    extern __shared__ __attribute__((aligned(16))) float red[];   // [R][C][2]
    const int b = blockIdx.y, s = blockIdx.x;
    const int C4 = C >> 2;
    const int per = (HW + nsplit - 1) / nsplit;
    const int p_begin = s * per, p_end = min(HW, p_begin + per);
    const float* xb = x + (size_t)b * HW * C;
    const int tid = threadIdx.x;
    if (C4 <= 256) {
        const int R = 256 / C4;
        const int c4 = tid % C4, r = tid / C4;
        float4 sm = make_float4(0.f, 0.f, 0.f, 0.f), sq = sm;
        if (r < R) {
            for (int p = p_begin + r; p < p_end; p += R) {
                const float4 v = *reinterpret_cast<const float4*>(xb + (size_t)p * C + 4 * c4);
                sm.x += v.x; sm.y += v.y; sm.z += v.z; sm.w += v.w;
                sq.x += v.x * v.x; sq.y += v.y * v.y; sq.z += v.z * v.z; sq.w += v.w * v.w;
            }
            float* d = red + ((size_t)r * C + 4 * c4) * 2;
            d[0] = sm.x; d[1] = sq.x; d[2] = sm.y; d[3] = sq.y; d[4] = sm.z; d[5] = sq.z; d[6] = sm.w; d[7] = sq.w;
        }
        __syncthreads();
        float* out = partial + ((size_t)(b * nsplit + s) * C) * 2;
        for (int i = tid; i < 2 * C; i += 256) {
            float acc = 0.f;
            for (int rr = 0; rr < R; ++rr) acc += red[(size_t)rr * 2 * C + i];
            out[i] = acc;
        }
    } else {
        float* out = partial + ((size_t)(b * nsplit + s) * C) * 2;
        for (int c4 = tid; c4 < C4; c4 += 256) {
            float4 sm = make_float4(0.f, 0.f, 0.f, 0.f), sq = sm;
            for (int p = p_begin; p < p_end; ++p) {
                const float4 v = *reinterpret_cast<const float4*>(xb + (size_t)p * C + 4 * c4);
                sm.x += v.x; sm.y += v.y; sm.z += v.z; sm.w += v.w;
                sq.x += v.x * v.x; sq.y += v.y * v.y; sq.z += v.z * v.z; sq.w += v.w * v.w;
            }
            float* d = out + 8 * c4;
            d[0] = sm.x; d[1] = sq.x; d[2] = sm.y; d[3] = sq.y; d[4] = sm.z; d[5] = sq.z; d[6] = sm.w; d[7] = sq.w;
        }
    }
}

struct CoefArgs {
    const float* part0; int nsplit0; int C0;
    const float* part1; int nsplit1; int C1;
    int B, HW, groups; float eps; int mode;
    const float* gamma; const float* beta; const float* ss; int ss_ld; const int* row;
    float* coef_a; float* coef_s;
    unsigned* bound_bits;      // optional: atomicMax of the bit pattern of the largest {sum of squares} entry seen
    unsigned* events;          // optional: sticky range-event word (EVC_RANGE_*), OR-ed, never cleared by a kernel
};

// fp16-split arithmetic (EVC_ARITH_F16X3) scales GroupNorm-ed operands by 8: they must stay below 65504 / 8
constexpr float F16_OPERAND_LIMIT = 65504.0f / 8.0f;
constexpr unsigned NAN_BITS = 0x7fc00000u;      // as a bound word: "the tensor holds a non-finite element"

__device__ __forceinline__ void raise_event(unsigned* events, unsigned bits) {
    // the word only ever gains bits: skip the atomic when they are already there (the common case is "nothing to report")
    if (events && (__hip_atomic_load(events, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bits) != bits) atomicOr(events, bits);
}

// grid (groups, B), block 256: the (split, channel) moments of one group are summed in double by the whole
// block (fixed assignment + fixed tree => deterministic), then the group's channels get their coefficients.
__global__ __launch_bounds__(256) void gn_coeffs_kernel(CoefArgs a) {
    __shared__ double red[2][4];
    __shared__ float redmx[4];
    __shared__ unsigned redbad[4];
    const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int C = a.C0 + a.C1;
    const int cpg = C / a.groups;
    const int c_begin = g * cpg;
    // the per-channel scale / shift of this thread's channel (cpg <= 256: one channel per thread) are fetched FIRST, so
    // their latency overlaps the moment loads below instead of following the reduction
    float pmul = 1.f, padd = 0.f;
    if (tid < cpg) {
        const int c = c_begin + tid;
        if (a.mode == 1) { pmul = a.gamma[c]; padd = a.beta[c]; }
        else if (a.mode == 2) {
            const float* r = a.ss + (size_t)(a.row ? a.row[b] : 0) * a.ss_ld;
            pmul = 1.f + r[c]; padd = r[C + c];
        }
    }
    double sm = 0.0, sq = 0.0;
    float mx = 0.f;            // largest sum-of-squares entry: sqrt(mx) bounds every element of the tensor(s)
    unsigned bad = 0;          // a moment that is not finite: the tensor holds a NaN / inf (fmaxf would drop a NaN silently)
    // channels of this group that live in source 0 / source 1
    const int n0 = max(0, min(c_begin + cpg, a.C0) - c_begin);
    const int n1 = cpg - n0;
    for (int i = tid; i < n0 * a.nsplit0; i += 256) {
        const int s = i / n0, cc = c_begin + (i - s * n0);
        const float2 e = *reinterpret_cast<const float2*>(a.part0 + ((size_t)(b * a.nsplit0 + s) * a.C0 + cc) * 2);
        sm += (double)e.x; sq += (double)e.y; mx = fmaxf(mx, e.y); bad |= !(fabsf(e.x) < 3.0e38f) | !(e.y < 3.0e38f);
    }
    if (n1 > 0) {
        const int c1 = c_begin + n0 - a.C0;
        for (int i = tid; i < n1 * a.nsplit1; i += 256) {
            const int s = i / n1, cc = c1 + (i - s * n1);
            const float2 e = *reinterpret_cast<const float2*>(a.part1 + ((size_t)(b * a.nsplit1 + s) * a.C1 + cc) * 2);
            sm += (double)e.x; sq += (double)e.y; mx = fmaxf(mx, e.y); bad |= !(fabsf(e.x) < 3.0e38f) | !(e.y < 3.0e38f);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sm += __shfl_xor(sm, off);
        sq += __shfl_xor(sq, off);
        mx = fmaxf(mx, __shfl_xor(mx, off));
    }
    bad = __any(bad) ? 1u : 0u;
    if ((tid & 63) == 0) { red[0][tid >> 6] = sm; red[1][tid >> 6] = sq; redmx[tid >> 6] = mx; redbad[tid >> 6] = bad; }
    __syncthreads();
    bad = redbad[0] | redbad[1] | redbad[2] | redbad[3];
    const float gmx = fmaxf(fmaxf(redmx[0], redmx[1]), fmaxf(redmx[2], redmx[3]));
    // max is order-independent, so the atomic keeps the result deterministic; non-negative floats order like their bits.
    // One atomic per block, and none when the word already holds a larger value (it only ever grows): 288 blocks hitting
    // one address cost the launch 2-3 us otherwise.
    // A non-finite moment turns the bound into the NaN pattern (it orders above every finite value and infinity): the
    // consumer's scale becomes NaN and its output with it -- a NaN / inf in the tensor is never turned into finite numbers.
    if (a.bound_bits && tid == 0) {
        const unsigned bits = bad ? NAN_BITS : __float_as_uint(gmx);
        if (bits > __hip_atomic_load(a.bound_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a.bound_bits, bits);
    }
    if (bad && tid == 0) raise_event(a.events, EVC_RANGE_NONFINITE);
    sm = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    sq = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    const double n = (double)cpg * (double)a.HW;
    const double mean = sm / n;
    double var = sq / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)a.eps));
    const float fmean = (float)mean;
    const float xmax = sqrtf(gmx);          // every element of this group's channels satisfies |x| <= xmax
    bool wide = false;
    float amax = 0.f, smax = 0.f;           // largest |mul| * rstd and |add| among this thread's channels
    for (int i = tid; i < cpg; i += 256) {
        const int c = c_begin + i;
        float mul = pmul, add = padd;
        if (i >= 256) {                       // (groups wider than 256 channels: not in this network)
            mul = 1.f; add = 0.f;
            if (a.mode == 1) { mul = a.gamma[c]; add = a.beta[c]; }
            else if (a.mode == 2) {
                const float* r = a.ss + (size_t)(a.row ? a.row[b] : 0) * a.ss_ld;
                mul = 1.f + r[c]; add = r[C + c];
            }
        }
        const float ca = rstd * mul;
        const float cs = add - fmean * ca;
        a.coef_a[(size_t)b * C + c] = ca;
        a.coef_s[(size_t)b * C + c] = cs;
        amax = fmaxf(amax, fabsf(ca)); smax = fmaxf(smax, fabsf(add));
        // x*ca + cs = (x - mean)*ca + add and |SiLU(v)| <= |v|: while |ca| (xmax + |mean|) + |add| stays below the fp16-split
        // kernels' operand limit no element of the normalised tensor can leave fp16's range (a sufficient condition, checked
        // on O(B*C) numbers here instead of on every element inside the convolution's K loop)
        wide |= !(fabsf(ca) * (xmax + fabsf(fmean)) + fabsf(add) < F16_OPERAND_LIMIT);
    }
    if (!a.events) return;                    // uniform
    if (!__syncthreads_or(wide)) return;
    if (bad) { if (tid == 0) raise_event(a.events, EVC_RANGE_NONFINITE); return; }
    // Rare path.  The cheap test failed (e.g. a near-constant group: rstd is huge, |x - mean| tiny): bound |x - mean| per
    // element by the deviation of its own moment entry, sum (x - mean)^2 = sumsq - 2 mean sum + n mean^2 over the entry's
    // pixels, n <= ceil(HW / nsplit) (a larger n only loosens the bound), plus a rounding allowance.
    float dev = 0.f;
    {
        const float n0f = (float)((a.HW + a.nsplit0 - 1) / a.nsplit0);
        for (int i = tid; i < n0 * a.nsplit0; i += 256) {
            const int s = i / n0, cc = c_begin + (i - s * n0);
            const float2 e = *reinterpret_cast<const float2*>(a.part0 + ((size_t)(b * a.nsplit0 + s) * a.C0 + cc) * 2);
            dev = fmaxf(dev, e.y - 2.f * fmean * e.x + n0f * fmean * fmean + 1e-5f * (e.y + n0f * fmean * fmean));
        }
        if (n1 > 0) {
            const int c1 = c_begin + n0 - a.C0;
            const float n1f = (float)((a.HW + a.nsplit1 - 1) / a.nsplit1);
            for (int i = tid; i < n1 * a.nsplit1; i += 256) {
                const int s = i / n1, cc = c1 + (i - s * n1);
                const float2 e = *reinterpret_cast<const float2*>(a.part1 + ((size_t)(b * a.nsplit1 + s) * a.C1 + cc) * 2);
                dev = fmaxf(dev, e.y - 2.f * fmean * e.x + n1f * fmean * fmean + 1e-5f * (e.y + n1f * fmean * fmean));
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        dev = fmaxf(dev, __shfl_xor(dev, off));
        amax = fmaxf(amax, __shfl_xor(amax, off));
        smax = fmaxf(smax, __shfl_xor(smax, off));
    }
    __shared__ float red3[3][4];
    if ((tid & 63) == 0) { red3[0][tid >> 6] = dev; red3[1][tid >> 6] = amax; red3[2][tid >> 6] = smax; }
    __syncthreads();
    if (tid == 0) {
        dev = fmaxf(fmaxf(red3[0][0], red3[0][1]), fmaxf(red3[0][2], red3[0][3]));
        amax = fmaxf(fmaxf(red3[1][0], red3[1][1]), fmaxf(red3[1][2], red3[1][3]));
        smax = fmaxf(fmaxf(red3[2][0], red3[2][1]), fmaxf(red3[2][2], red3[2][3]));
        if (!(amax * sqrtf(fmaxf(dev, 0.f)) + smax < F16_OPERAND_LIMIT)) raise_event(a.events, EVC_RANGE_F16_OPERAND);
    }
}

// max over channel ranges of a moments tensor [B][nsplit][C][2]: the bound of a tensor that has no GroupNorm in
// front of it (evc_moments_bound_f32).  grid (nsplit, B, ranges): range z = channels [c_begin + z*c_count, +c_count).
__global__ __launch_bounds__(256) void moments_bound_kernel(const float* __restrict__ part, int nsplit, int C, int c_begin,
                                                            int c_count, unsigned* __restrict__ bound_bits,
                                                            unsigned* __restrict__ events) {
    const float* row = part + ((size_t)(blockIdx.y * nsplit + blockIdx.x) * C + c_begin + blockIdx.z * c_count) * 2;
    bound_bits += blockIdx.z;
    float mx = 0.f;
    unsigned bad = 0;
    for (int i = threadIdx.x; i < c_count; i += 256) {
        const float2 e = *reinterpret_cast<const float2*>(row + 2 * i);
        mx = fmaxf(mx, e.y);
        bad |= !(fabsf(e.x) < 3.0e38f) | !(e.y < 3.0e38f);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    bad = __any(bad) ? 1u : 0u;
    __shared__ float redmx[4];
    __shared__ unsigned redbad[4];
    if ((threadIdx.x & 63) == 0) { redmx[threadIdx.x >> 6] = mx; redbad[threadIdx.x >> 6] = bad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        bad = redbad[0] | redbad[1] | redbad[2] | redbad[3];
        const unsigned bits = bad ? NAN_BITS : __float_as_uint(fmaxf(fmaxf(redmx[0], redmx[1]), fmaxf(redmx[2], redmx[3])));
        if (bits > __hip_atomic_load(bound_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(bound_bits, bits);
        if (bad) raise_event(events, EVC_RANGE_NONFINITE);
    }
}

__device__ __forceinline__ float act_fn(float v, int act) {
    if (act == EVC_ACT_SILU) return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
    if (act == EVC_ACT_RELU) return fmaxf(v, 0.0f);
    return v;
}

__global__ void affine_act_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ ca,
                                  const float* __restrict__ cs, int act, int HW, int C, int ld_coef, int ld_out,
                                  size_t total4) {
    const int C4 = C >> 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const size_t pix = i / C4;
        const int b = (int)(pix / HW);
        float4 v = reinterpret_cast<const float4*>(x)[i];
        if (ca) {
            const float4 a = *reinterpret_cast<const float4*>(ca + (size_t)b * ld_coef + 4 * c4);
            const float4 s = *reinterpret_cast<const float4*>(cs + (size_t)b * ld_coef + 4 * c4);
            v.x = v.x * a.x + s.x; v.y = v.y * a.y + s.y; v.z = v.z * a.z + s.z; v.w = v.w * a.w + s.w;
        }
        v.x = act_fn(v.x, act); v.y = act_fn(v.y, act); v.z = act_fn(v.z, act); v.w = act_fn(v.w, act);
        *reinterpret_cast<float4*>(y + pix * (size_t)ld_out + 4 * c4) = v;
    }
}

// SPADE act-norm (reference models/better/layerspp.py:152-173 MySPADE + :518-549 get_act_norm, norm == 'spade'):
//   y = act( [ (x*a + s) * G + Bt ] * (1 + scale) + shift )
// a, s: parameter-free GroupNorm coefficients per (sample, channel); G = 1 + gamma(cond), Bt = beta(cond): per-pixel maps
// (they depend on the conditioning frames only, the host computes them once per chunk); scale / shift: the AdaGN row of
// the sample's step label (absent for the network's final norm).
__global__ void spade_act_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ ca,
                                 const float* __restrict__ cs, int ld_coef, const float* __restrict__ gmap,
                                 const float* __restrict__ bmap, int ld_map, const float* __restrict__ ss_scale,
                                 const float* __restrict__ ss_shift, int ld_ss, const int* __restrict__ row, int act,
                                 int HW, int C, int ld_out, size_t total4) {
    const int C4 = C >> 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const size_t pix = i / C4;
        const int b = (int)(pix / HW);
        float4 v = reinterpret_cast<const float4*>(x)[i];
        const float4 a = *reinterpret_cast<const float4*>(ca + (size_t)b * ld_coef + 4 * c4);
        const float4 s = *reinterpret_cast<const float4*>(cs + (size_t)b * ld_coef + 4 * c4);
        const float4 g = *reinterpret_cast<const float4*>(gmap + pix * (size_t)ld_map + 4 * c4);
        const float4 t = *reinterpret_cast<const float4*>(bmap + pix * (size_t)ld_map + 4 * c4);
        v.x = (v.x * a.x + s.x) * g.x + t.x; v.y = (v.y * a.y + s.y) * g.y + t.y;
        v.z = (v.z * a.z + s.z) * g.z + t.z; v.w = (v.w * a.w + s.w) * g.w + t.w;
        if (ss_scale) {
            const size_t r = (size_t)(row ? row[b] : 0) * ld_ss + 4 * c4;
            const float4 sc = *reinterpret_cast<const float4*>(ss_scale + r);
            const float4 sh = *reinterpret_cast<const float4*>(ss_shift + r);
            v.x = v.x * (1.0f + sc.x) + sh.x; v.y = v.y * (1.0f + sc.y) + sh.y;
            v.z = v.z * (1.0f + sc.z) + sh.z; v.w = v.w * (1.0f + sc.w) + sh.w;
        }
        v.x = act_fn(v.x, act); v.y = act_fn(v.y, act); v.z = act_fn(v.z, act); v.w = act_fn(v.w, act);
        *reinterpret_cast<float4*>(y + pix * (size_t)ld_out + 4 * c4) = v;
    }
}

}  // namespace

extern "C" int evc_spade_act_nhwc_f32(const float* x, float* y, const float* coef_a, const float* coef_s, int ld_coef,
                                      const float* gmap, const float* bmap, int ld_map, const float* ss_scale,
                                      const float* ss_shift, int ld_ss, const int* row, int act, int B, int HW, int C,
                                      int ld_out, void* stream) {
    if (!x || !y || !coef_a || !coef_s || !gmap || !bmap || B <= 0 || HW <= 0 || C <= 0 || (C & 3)) return EVC_EINVAL;
    if ((ss_scale == nullptr) != (ss_shift == nullptr)) return EVC_EINVAL;
    if (ld_coef < C || ld_map < C || ld_out < C || ((ld_coef | ld_map | ld_out) & 3)) return EVC_EINVAL;
    if (ss_scale && (ld_ss < C || (ld_ss & 3))) return EVC_EINVAL;
    const size_t total4 = (size_t)B * HW * (C >> 2);
    const int grid = (int)((total4 + 255) / 256 > 4096 ? 4096 : (total4 + 255) / 256);
    hipLaunchKernelGGL(spade_act_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, y, coef_a, coef_s, ld_coef,
                       gmap, bmap, ld_map, ss_scale, ss_shift, ld_ss, row, act, HW, C, ld_out, total4);
    return hipGetLastError() == hipSuccess ? EVC_OK : EVC_ELAUNCH;
}

extern "C" int evc_chan_stats_f32(const float* x, float* partial, int B, int HW, int C, int nsplit, void* stream) {
    if (!x || !partial || B <= 0 || HW <= 0 || C <= 0 || (C & 3) || nsplit <= 0 || nsplit > HW) return EVC_EINVAL;
    const int C4 = C >> 2;
    const int R = C4 <= 256 ? 256 / C4 : 0;
    const size_t lds = (size_t)R * C * 2 * sizeof(float);
    if (lds > 64 * 1024) return EVC_EUNSUPPORTED;
    hipLaunchKernelGGL(chan_stats_kernel, dim3(nsplit, B), dim3(256), lds, (hipStream_t)stream, x, partial, HW, C,
                       nsplit);
    return hipGetLastError() == hipSuccess ? EVC_OK : EVC_ELAUNCH;
}

extern "C" int evc_gn_coeffs_bound_f32(const float* part0, int nsplit0, int C0, const float* part1, int nsplit1, int C1,
                                       int B, int HW, int groups, float eps, int mode, const float* gamma,
                                       const float* beta, const float* ss, int ss_ld, const int* row, float* coef_a,
                                       float* coef_s, unsigned* bound_bits, unsigned* events, void* stream);

extern "C" int evc_gn_coeffs_f32(const float* part0, int nsplit0, int C0, const float* part1, int nsplit1, int C1,
                                 int B, int HW, int groups, float eps, int mode, const float* gamma,
                                 const float* beta, const float* ss, int ss_ld, const int* row, float* coef_a,
                                 float* coef_s, void* stream) {
    return evc_gn_coeffs_bound_f32(part0, nsplit0, C0, part1, nsplit1, C1, B, HW, groups, eps, mode, gamma, beta, ss,
                                   ss_ld, row, coef_a, coef_s, nullptr, nullptr, stream);
}

extern "C" int evc_moments_bound_f32(const float* part, int nsplit, int C, int c_begin, int c_count, int n_ranges, int B,
                                     unsigned* bound_bits, unsigned* events, void* stream) {
    if (!part || !bound_bits || nsplit <= 0 || C <= 0 || B <= 0 || c_begin < 0 || c_count <= 0 || n_ranges <= 0 ||
        c_begin + (long long)n_ranges * c_count > C)
        return EVC_EINVAL;
    hipLaunchKernelGGL(moments_bound_kernel, dim3(nsplit, B, n_ranges), dim3(256), 0, (hipStream_t)stream, part, nsplit, C,
                       c_begin, c_count, bound_bits, events);
    return hipGetLastError() == hipSuccess ? EVC_OK : EVC_ELAUNCH;
}

extern "C" int evc_gn_coeffs_bound_f32(const float* part0, int nsplit0, int C0, const float* part1, int nsplit1, int C1,
                                       int B, int HW, int groups, float eps, int mode, const float* gamma,
                                       const float* beta, const float* ss, int ss_ld, const int* row, float* coef_a,
                                       float* coef_s, unsigned* bound_bits, unsigned* events, void* stream) {
    if (!part0 || C0 <= 0 || nsplit0 <= 0 || C1 < 0 || (C1 > 0 && (!part1 || nsplit1 <= 0))) return EVC_EINVAL;
    if (B <= 0 || HW <= 0 || groups <= 0 || (C0 + C1) % groups != 0 || !coef_a || !coef_s) return EVC_EINVAL;
    if (mode < 0 || mode > 2 || (mode == 1 && (!gamma || !beta)) || (mode == 2 && (!ss || ss_ld < 2 * (C0 + C1))))
        return EVC_EINVAL;
    CoefArgs a{part0, nsplit0, C0, part1, nsplit1, C1, B, HW, groups, eps, mode, gamma, beta, ss, ss_ld, row,
               coef_a, coef_s, bound_bits, events};
    hipLaunchKernelGGL(gn_coeffs_kernel, dim3(groups, B), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? EVC_OK : EVC_ELAUNCH;
}

extern "C" int evc_affine_act_nhwc_f32(const float* x, float* y, const float* coef_a, const float* coef_s, int act,
                                       int B, int HW, int C, int ld_coef, int ld_out, void* stream) {
    if (!x || !y || B <= 0 || HW <= 0 || C <= 0 || (C & 3) || ((coef_a == nullptr) != (coef_s == nullptr)))
        return EVC_EINVAL;
    if (ld_coef == 0) ld_coef = C;
    if (ld_out == 0) ld_out = C;
    if (ld_coef < C || ld_out < C || (ld_coef & 3) || (ld_out & 3)) return EVC_EINVAL;
    const size_t total4 = (size_t)B * HW * (C >> 2);
    int grid = (int)((total4 + 255) / 256 > 2048 ? 2048 : (total4 + 255) / 256);
    hipLaunchKernelGGL(affine_act_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, y, coef_a, coef_s, act,
                       HW, C, ld_coef, ld_out, total4);
    return hipGetLastError() == hipSuccess ? EVC_OK : EVC_ELAUNCH;
}
