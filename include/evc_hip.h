/* evc_hip.h -- C ABI of libevc_hip.so: the MI355X (gfx950) kernels of the decode hot path.
 *
 * Boundary rules: plain pointers and sizes, no torch types, the caller owns every
 * buffer (no allocation inside), every call is stream-ordered on the `stream` argument
 * (a hipStream_t passed as void*), returns 0 on success and a negative EVC_E* code otherwise,
 * never throws.  All tensors are float32.  "NHWC" = [B][H][W][C] contiguous, C fastest.
 *
 * What each entry point replaces in the reference is cited next to it (paths relative to the
 * reference repository root).
 */
#ifndef EVC_HIP_H
#define EVC_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define EVC_OK 0
#define EVC_EINVAL (-1)   /* bad shape / null pointer / misaligned channel count */
#define EVC_EUNSUPPORTED (-2)
#define EVC_ELAUNCH (-3)  /* hipGetLastError() != hipSuccess after the launch */

#define EVC_ACT_NONE 0
#define EVC_ACT_SILU 1
#define EVC_ACT_RELU 2

/* How the convolution multiplies (both are fp32 convolutions with fp32 accumulation):
 *   EVC_ARITH_F32     v_mfma_f32_32x32x2_f32: exact fp32 products, k-ordered fmaf chain;
 *   EVC_ARITH_BF16X6  every fp32 operand is split EXACTLY into three bf16 values and the six significant cross
 *                     products run on v_mfma_f32_32x32x16_bf16 (16x the f32 MFMA rate); the dropped terms are below
 *                     one fp32 rounding of a product.  Measured error against fp64 on MI355X: equal to or below
 *                     that of EVC_ARITH_F32 (tools/split_numerics.hip). */
#define EVC_ARITH_F32 0
#define EVC_ARITH_BF16X6 1
/*   EVC_ARITH_F16X3   both operands are scaled by powers of two into fp16's range (weights per tensor at pack time,
 *                     activations by 8 -- nothing is clamped: an element beyond fp16's range comes out as NaN and the
 *                     EVC_RANGE_* word below says beforehand whether one can exist), split 2-way into fp16 (11 + 11 significand bits,
 *                     2^-22 relative) and three cross products run on v_mfma_f32_32x32x16_f16 (bf16 rate); the exact
 *                     inverse scale is applied to the fp32 accumulator.  Half the MFMAs and two thirds of the LDS
 *                     bytes of BF16X6.  Measured error against fp64: BELOW both F32 and BF16X6 for O(1) operands
 *                     (profiles/r02_split_numerics.log) -- fewer accumulator roundings -- but operands spanning more
 *                     than fp16's exponent range are NOT representable: use it only where the input is
 *                     GroupNorm-normalised / activated (the caller's choice, made when packing the weights). */
#define EVC_ARITH_F16X3 2

/* Library / device identification. evc_arch() returns the gfx target the code object was built
 * for ("gfx950"). evc_device_ok() returns 1 when the current HIP device can run it. */
const char* evc_version(void);
const char* evc_arch(void);
int evc_device_ok(void);

/* Measurement aid (bench.py): a one-wave kernel on `stream` that idles for at most spin_us microseconds (<= 10 s), or until
 * the device word *stop (may be NULL) becomes non-zero -- the caller writes it stream-ordered behind the work it wants
 * characterised, so the probe never outlives that work -- and writes out2[0] = shader-clock ticks (s_memtime), out2[1] =
 * 100 MHz reference ticks (s_memrealtime) elapsed meanwhile; out2 is device or pinned host memory.  out2[0] / out2[1] / 10
 * = the shader clock in GHz the chip held while other streams ran -- the convolution kernels run the package into its
 * power cap, so this is what the roofline's clock is. */
int evc_clock_probe(unsigned long long* out2, int spin_us, const unsigned* stop, void* stream);

/* ---- upfirdn2d: the reference's own native op ---------------------------------------------
 * Replaces pybind `upfirdn2d(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0,
 * pad_y1)` (models/better/op/upfirdn2d.cpp:12-23, upfirdn2d_kernel.cu:209-243).  `input` is the
 * reference's (major = N*C, in_h, in_w, minor = 1) view, i.e. NCHW planes; `out` must hold
 * major*out_h*out_w floats with out_h = (in_h*up_y + pad_y0 + pad_y1 - kh)/down_y + 1 (same for w).
 * The FIR kernel (kh*kw <= 64 taps) is a HOST pointer (it is 16 floats in this workload). */
int evc_upfirdn2d_f32(const float* input, float* out, const float* kernel_host, int major, int in_h, int in_w,
                      int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1,
                      int pad_y0, int pad_y1, void* stream);

/* Same maths on an NHWC tensor, with an optional per-(b,c) affine + activation applied to every
 * input sample before filtering (zero padding stays zero).  Fuses get_act_norm + upsample_2d /
 * downsample_2d of ResnetBlockBigGANppGN.forward (models/better/layerspp.py:596-611). */
int evc_upfirdn2d_nhwc_f32(const float* x, float* out, const float* kernel_host, int B, int H, int W, int C,
                           int kh, int kw, int up, int down, int pad0, int pad1, const float* coef_a,
                           const float* coef_s, int act, void* stream);

/* ---- layout -------------------------------------------------------------------------------
 * Pack up to two NCHW tensors (x: C0 planes, cond: C1 planes) into one zero-padded NHWC tensor with
 * Cpad channels: torch.cat([x, cond], 1) of NCSNpp.forward (models/better/ncsnpp_more.py:256-257). */
int evc_pack_nchw_to_nhwc_f32(const float* x0, int C0, const float* x1, int C1, float* out, int Cpad, int B,
                              int H, int W, void* stream);
/* out[b][c][h][w] = in[b][h][w][c] for c < C; `ld` is the NHWC channel stride of `in`. */
int evc_nhwc_to_nchw_f32(const float* in, int ld, float* out, int B, int C, int H, int W, void* stream);

/* ---- GroupNorm ----------------------------------------------------------------------------
 * Per-channel partial moments of an NHWC tensor: partial[b][s][c] = {sum, sum of squares} over the
 * s-th of `nsplit` pixel ranges.  First half of nn.GroupNorm (models/better/layerspp.py:473-477). */
int evc_chan_stats_f32(const float* x, float* partial, int B, int HW, int C, int nsplit, void* stream);

/* Turn moments into per-(b,c) affine coefficients so that  norm(x) = x*coef_a + coef_s :
 *   mode 0: plain group norm;  mode 1: * gamma + beta (AttnBlockpp.GroupNorm_0, final Norm_0);
 *   mode 2: * (1 + scale) + shift, the AdaGN of get_act_norm.forward
 *           (models/better/layerspp.py:518-549); scale = ss[row[b]*ss_ld + c], shift = ...[C + c].
 * Channels come from up to two tensors (virtual concat, C = C0 + C1); groups may straddle them. */
int evc_gn_coeffs_f32(const float* part0, int nsplit0, int C0, const float* part1, int nsplit1, int C1, int B,
                      int HW, int groups, float eps, int mode, const float* gamma, const float* beta,
                      const float* ss, int ss_ld, const int* row, float* coef_a, float* coef_s, void* stream);

/* Range events of the fp16-split arithmetic (EVC_ARITH_F16X3).  That arithmetic scales its operands by powers of two
 * into fp16's range WITHOUT clamping: an operand element beyond the range becomes inf in the split and NaN in the output
 * (loud, never a silently saturated number), and a NaN / inf input stays non-finite.  Whether that can happen is decided
 * on O(B*C) numbers by the kernels that see every tensor's moments -- the two functions below -- which OR these bits into
 * the caller's sticky device word `events` (may be NULL; never cleared by a kernel):
 *   EVC_RANGE_NONFINITE    a tensor's moments are not finite: it holds a NaN or an infinity;
 *   EVC_RANGE_F16_OPERAND  for some channel |coef_a| * max|x| + |coef_s| >= 65504 / 8: a GroupNorm-ed (and activated,
 *                          |SiLU(v)| <= |v|) operand element MAY leave fp16's range.  Zero means none can (a sufficient
 *                          condition).  Remedy: EVC_ARITH_BF16X6 for that model (host: EVC_CONV_ARITH=bf16x6). */
#define EVC_RANGE_NONFINITE 1u
#define EVC_RANGE_F16_OPERAND 2u

/* evc_gn_coeffs_f32 that additionally raises *bound_bits (atomic max; the caller zeroes it first) to the bit pattern of
 * the largest {sum of squares} entry among the moments it read: sqrt of that float bounds every element of the tensor(s),
 * because each element belongs to exactly one entry (the NaN pattern 0x7fc00000 when a moment is not finite: consumers
 * then scale by NaN).  Deterministic (max is order-independent).  bound_bits and events may be NULL. */
int evc_gn_coeffs_bound_f32(const float* part0, int nsplit0, int C0, const float* part1, int nsplit1, int C1, int B,
                            int HW, int groups, float eps, int mode, const float* gamma, const float* beta,
                            const float* ss, int ss_ld, const int* row, float* coef_a, float* coef_s,
                            unsigned* bound_bits, unsigned* events, void* stream);
/* The same bound over n_ranges consecutive channel ranges [c_begin + z*c_count, + c_count) of one moments tensor
 * [B][nsplit][C][2], range z into bound_bits[z] (for tensors that are not followed by a GroupNorm: the q | k | v
 * projection feeding attention has three). */
int evc_moments_bound_f32(const float* part, int nsplit, int C, int c_begin, int c_count, int n_ranges, int B,
                          unsigned* bound_bits, unsigned* events, void* stream);

/* y = act(x*coef_a[b][c] + coef_s[b][c]) elementwise on NHWC: the stand-alone form of the fused load.  One pass per
 * tensor instead of once per filter tap inside the convolution (SiLU costs MFMA issue slots there; HBM is cheap).
 * `coef_*` rows have stride ld_coef (0 = C) so a slice of a concat's coefficients can be used; `y` rows have stride
 * ld_out (0 = C) so two tensors can be activated side by side into one buffer (materialised concat). */
int evc_affine_act_nhwc_f32(const float* x, float* y, const float* coef_a, const float* coef_s, int act, int B,
                            int HW, int C, int ld_coef, int ld_out, void* stream);

/* SPADE act-norm of the conditioning-by-normalisation variant of the score network (reference
 * models/better/layerspp.py:152-173 MySPADE.forward + :518-549 get_act_norm.forward with norm == 'spade'):
 *   y = act( [ (x*coef_a[b][c] + coef_s[b][c]) * gmap[pixel][c] + bmap[pixel][c] ] * (1 + scale[row[b]][c]) + shift[row[b]][c] )
 * coef_*: parameter-free GroupNorm coefficients (evc_gn_coeffs_f32 mode 0, eps 1e-6); gmap = 1 + gamma(cond) and
 * bmap = beta(cond): NHWC maps with row stride ld_map (they depend on the conditioning frames only); ss_scale / ss_shift:
 * AdaGN table columns with row stride ld_ss, NULL for the final norm (no time embedding); `row` NULL = row 0.  x is
 * contiguous [B*HW][C]; all pointers may be offset to a channel slice of wider buffers (ld_* are the full widths). */
int evc_spade_act_nhwc_f32(const float* x, float* y, const float* coef_a, const float* coef_s, int ld_coef,
                           const float* gmap, const float* bmap, int ld_map, const float* ss_scale,
                           const float* ss_shift, int ld_ss, const int* row, int act, int B, int HW, int C, int ld_out,
                           void* stream);

/* ---- convolution as implicit GEMM on the matrix cores -----------------------------------------
 * Two arithmetics, both fp32 in / fp32 accumulate (see EVC_ARITH_* above); the packing of the weights selects one:
 *   EVC_ARITH_BF16X6 (default of the Python host): conv_split_rr_kernel for 3x3 filters on tiles made of whole image
 *     rows (W divides 128: every 3x3 layer of the score network and of ELIC's 8..128-wide stages), conv_split_kernel
 *     for everything else (1x1, 5x5, odd widths);
 *   EVC_ARITH_F32: conv_igemm_kernel (v_mfma_f32_32x32x2_f32).
 * Stride-1 "same" convolution (odd KH x KW, zero padding) over the virtual concat [src0 | src1] of
 * NHWC tensors.  Replaces nn.Conv2d 3x3 / 1x1 (models/better/layers.py:89-113), NIN
 * (models/better/layers.py:535-544), nn.Linear (time-embedding MLP, Dense_0) and the ELIC conv
 * stacks (Network.py:106-166).
 *   in   = act_in(src * coef_a + coef_s)            (coef may be NULL; zero padding after act)
 *   out  = act_out((conv(in, w) + bias + res) * out_scale)
 * Weights are pre-packed by evc_conv_pack_weights_f32 into [KH*KW][Ci/16][CoPad][16].
 * C0, C1 must be multiples of 16.  `ws` is a workspace of evc_conv_workspace_bytes() bytes
 * (split-K partial sums; may be NULL when that returns 0). */
typedef struct {
    const float* src0; const float* src1; int C0; int C1;
    int ld0; int ld1;      /* row strides (floats) of src0 / src1; 0 = C0 / C1 (dense). Lets a source be a
                              channel slice of a wider NHWC tensor. Must be multiples of 4. */
    const float* coef_a; const float* coef_s; int act_in;
    const float* w_packed; const float* bias; const float* res; int ld_res;
    float out_scale; int act_out;
    float* out; int ld_out;
    int B; int H; int W; int Co; int KH; int KW;
    int splits;            /* 0 = choose automatically */
    float* stats_out;      /* optional: per-channel moments of `out`, fused into the epilogue, in the layout of
                              evc_chan_stats_f32 with nsplit = evc_conv_stats_splits() ([B][nsplit][Co][2] = {sum, sumsq}
                              of each pixel run). Only honoured when that is > 0; else must be NULL. */
    int arith;             /* EVC_ARITH_*: must match the packing of w_packed */
    const unsigned* in_bound;  /* EVC_ARITH_F16X3 only, optional (NULL otherwise): device word holding the bit pattern of a
                              float S such that every element of src0 / src1 satisfies |x| <= sqrt(S) -- written by
                              evc_gn_coeffs_bound_f32 / evc_moments_bound_f32 from the tensors' moments.  The kernel scales
                              the sources by the power of two that brings sqrt(S) into [64, 128) before the fp16 split and
                              the accumulator by its inverse: lets the fp16 arithmetic take inputs that no GroupNorm has
                              normalised (1x1 skip convolutions, NIN output projections). */
    /* Optional FUSED 1x1 OPERAND (x2_w_packed != NULL; only where evc_conv_fused_1x1_supported() says so: EVC_ARITH_F16X3,
     * 3x3 filters on the row-reuse kernel):
     *     out = act_out((conv(act_in(cat[src0,src1]*a+s), w) + conv1x1(cat[x2_src0,x2_src1], x2_w) + bias + res) * out_scale)
     * = a res-block's Conv_1 plus its 1x1 skip convolution Conv_2 on the block input (reference
     * models/better/layerspp.py:603-624: x = Conv_2(x); return (x + h) / sqrt(2)) in ONE launch and one accumulator:
     * x2's K-steps run first, the accumulators are rescaled by the (power-of-two) ratio of the two operands' scales, the
     * 3x3 K loop continues into them.  x2 is a raw tensor of the output's B x H x W (two sources like src, row strides
     * x2_ld*, 0 = C), x2_w_packed its [Co][x2_C0+x2_C1][1][1] weights packed for EVC_ARITH_F16X3, x2_bound its element
     * bound (as in_bound, required).  `bias` must already hold the sum of both convolutions' biases. */
    const float* x2_src0; const float* x2_src1; int x2_C0; int x2_C1; int x2_ld0; int x2_ld1;
    const float* x2_w_packed; const unsigned* x2_bound;
} evc_conv_args;
int evc_conv_co_pad(int Co);
long long evc_conv_packed_floats(int Co, int Ci, int KH, int KW);
/* w: [Co][Ci][KH][KW] (PyTorch Conv2d layout, device) -> packed (device). */
int evc_conv_pack_weights_f32(const float* w, float* packed, int Co, int Ci, int KH, int KW, void* stream);
/* The same for any arithmetic: EVC_ARITH_F32 -> the layout above (4 bytes per element), EVC_ARITH_BF16X6 ->
 * [KH*KW][Ci/16][3 planes][CoPad][16 bf16] (6 bytes per element), EVC_ARITH_F16X3 -> a 256-byte header (inverse
 * accumulator scale, weight scale) + [KH*KW][Ci/16][2 planes][CoPad][16 fp16] (4 bytes per element). */
long long evc_conv_packed_bytes(int Co, int Ci, int KH, int KW, int arith);
int evc_conv_pack_weights(const float* w, void* packed, int Co, int Ci, int KH, int KW, int arith, void* stream);
/* Process-wide tuning switches of the convolution dispatch (A/B measurements, tests of non-default kernels; results are
 * the same up to fp32 summation order): "wide_tiles" (default 1: 256-pixel row tiles on large unsplit grids), "row_reuse"
 * (default 1: the row-reuse kernel for 3x3 filters), "tail_split" (default 1: K-split tail of 1.x / 2.x-round grids), "wide256"
 * (default 1: conv_wide_kernel, the 256 x 192 one-workgroup-per-CU f16x3 kernel, on grids of at least one full round), "wide_mid"
 * (default 1: the same kernel with a uniform K split on grids below one round), "wide_cut" (default 1: on grids of 129..255 such
 * workgroups every tile is cut into a long and a short K piece so that the idle CUs take the short ones).
 * Returns EVC_EINVAL for an unknown name. */
int evc_conv_set_option(const char* name, int value);
/* Name of the kernel template instance evc_conv2d_nhwc_f32 launches for these arguments, as rocprofv3 --kernel-trace --stats
 * prints it (e.g. "conv_wide_kernel<2, 4, false>"): lets a profile be matched to launches without mirroring the dispatch.
 * Returns the string length (truncated to n - 1) or EVC_EINVAL. */
int evc_conv_kernel_name(const evc_conv_args* a, char* buf, int n);
int evc_conv_choose_splits(const evc_conv_args* a);
int evc_conv_fused_1x1_supported(const evc_conv_args* a);   /* 1: these arguments may carry the fused 1x1 operand */
/* The number of pixel runs per image (H*W/64 or H*W/32) for which the fused moments will be written, when they are
 * available for these arguments (H*W % 64 == 0 and either split-K -- the combine kernel writes them -- or only full
 * tiles), else 0: the caller then runs evc_chan_stats_f32 on the output instead. */
int evc_conv_stats_splits(const evc_conv_args* a);
long long evc_conv_workspace_bytes(const evc_conv_args* a);
int evc_conv2d_nhwc_f32(const evc_conv_args* a, float* ws, void* stream);
/* Measurement hook: the same launch with two hipEvent_t handles (either may be NULL) recorded on `stream` immediately before
 * the convolution kernel and immediately after it, i.e. before the split-K combine kernel: the interval is the convolution
 * kernel's own duration, the figure rocprofv3 --kernel-trace reports for it (bench.py's roofline leg). */
int evc_conv2d_nhwc_profiled_f32(const evc_conv_args* a, float* ws, void* stream, void* ev_start, void* ev_conv_end);

/* ---- multi-head spatial self-attention ------------------------------------------------------
 * out[b][n][h*D + d] = sum_m softmax_m(q[b][n][h].k[b][m][h] * scale) * v[b][m][h][d]
 * q, k, v: [B][N][ld_*] token-major with head h at channel offset h*D; D in {32, 64, 128, 192, 256}
 * (256: always the f32 kernel -- the fp16-split kernel keeps Q and O in registers, D/2 + D/2 of them).
 * Replaces the two einsums + softmax of AttnBlockpp.forward (models/better/layerspp.py:241-246). */
int evc_attention_f32(const float* q, const float* k, const float* v, int ld_qkv, float* out, int ld_out, int B,
                      int heads, int N, int D, float scale, void* stream);
/* The same with a workspace of evc_attention_workspace_bytes() bytes (0 = none needed): launches that would leave
 * SIMDs idle split the KEY range over several workgroups (each leaves an unnormalised partial + running max / sum)
 * and a merge kernel applies the exact online-softmax combination.  Results equal evc_attention_f32 up to fp32
 * rounding of the merge.  The workspace also holds the fp16 kernel's K / V tile images (evc_attention_set_option), so it is
 * non-zero whenever N >= 128 and D is one of 32 / 64 / 128 / 192. */
long long evc_attention_workspace_bytes(int B, int heads, int N, int D);
/* A/B switch: "kv_planes" (default 1) -- evc_attention_f16x3_f32 with >= 512 keys converts K / V ONCE per launch into per-tile
 * images of its LDS layout (a pre-pass into the workspace) and stages them by LDS-DMA, instead of converting every 32-key tile
 * again in every query block.  Same numbers either way.  "f16_min_keys" (default 128): fewest keys for which
 * evc_attention_f16x3_f32 runs the fp16-split kernel (below it the f32-MFMA kernel: B = 9, 64 keys, 4 heads of 192: 26 us vs 40 us).
 * Returns EVC_EINVAL for an unknown name. */
int evc_attention_set_option(const char* name, int value);
int evc_attention_ws_f32(const float* q, const float* k, const float* v, int ld_qkv, float* out, int ld_out, int B,
                         int heads, int N, int D, float scale, float* ws, void* stream);

/* The same attention on the fp16 matrix cores (EVC_ARITH_F16X3 scheme: operands scaled by powers of two, 2-way fp16
 * split, three v_mfma_f32_32x32x16_f16 per product, fp32 accumulation and fp32 softmax).  `bounds` = three device words
 * holding the bit patterns of floats Sq, Sk, Sv with |q| <= sqrt(Sq), |k| <= sqrt(Sk), |v| <= sqrt(Sv) elementwise
 * (evc_moments_bound_f32 over the q / k / v channel ranges of the projection's moments).  Same workspace as
 * evc_attention_ws_f32 (pass NULL when evc_attention_workspace_bytes() is 0). */
int evc_attention_f16x3_f32(const float* q, const float* k, const float* v, int ld_qkv, float* out, int ld_out, int B,
                            int heads, int N, int D, float scale, const unsigned* bounds, float* ws, void* stream);

/* ---- frame axis (pseudo-3-D score network, config.model.arch = unetmorepseudo3d) -----------------
 * Video activations are x[b][n][pixel][c]: an NHWC tensor of B*N images, the N frames of a sample adjacent.  A per-frame Conv2d
 * is then evc_conv2d_nhwc_f32 over B*N images; PseudoConv3d's Conv1d over the frames (reference models/better/layers3d.py:274,
 * 294-297) is evc_conv2d_nhwc_f32 with KH = 3 (or 1), KW = 1 over B "images" of N rows x H*W columns; the 3-D GroupNorm's
 * moments are evc_chan_stats_f32's with N times the pixel runs.  The operations below have no 2-D counterpart. */
/* nn.GroupNorm of AttnBlockpp1d on (B*H*W, C, N) (layers3d.py:89-90,107): per pixel, moments over (C / groups channels x N
 * frames), biased variance, then * gamma[c] + beta[c].  y may alias x. */
int evc_frame_group_norm_f32(const float* x, float* y, const float* gamma, const float* beta, int B, int N, int HW, int C,
                             int groups, float eps, void* stream);
/* Attention over the N <= 8 frames of one pixel (AttnBlockpp1d.forward, layers3d.py:112-118): q | k | v are channel ranges
 * [0, C) | [C, 2C) | [2C, 3C) of qkv rows of stride ld_qkv; per (pixel, head) softmax_i(q_t . k_i * scale) applied to v. */
int evc_frame_attention_f32(const float* qkv, int ld_qkv, float* out, int ld_out, int B, int N, int HW, int C, int heads,
                            float scale, void* stream);
/* The 1x1 "converter" convolution over the frame axis (reference models/better/ncsnpp_more.py:213-216,226-228,328-335,
 * 344-351): y[b][m][j] = sum_n w[m][n] * x[b][n][j] + bias[m], j < inner = H*W*C (multiple of 4), N, M <= 8. */
int evc_frame_mix_f32(const float* x, float* y, const float* w, const float* bias, int B, int N, int M, long long inner,
                      void* stream);

/* The temporal taps of nn.Conv3d (config.model.arch = unetmore3d, reference models/better/layers3d.py:225-243) side by side
 * along the channels: y[b][n][pixel][kt*C + c] = x[b][n + kt - 1][pixel][c], zeros beyond a sample's first / last frame,
 * kt = 0, 1, 2.  The 3 x 3 x 3 convolution is then evc_conv2d_nhwc_f32 (KH = KW = 3) over B*N images with 3C input channels
 * and the weight (Co, Ci, kt, kh, kw) read as (Co, kt*Ci + ci, kh, kw).  x must already be activated (zeros stay zeros). */
int evc_frame_taps_f32(const float* x, float* y, int B, int N, int HW, int C, void* stream);

/* ---- sampler steps (elementwise, flat over n floats) ---------------------------------------- */
/* DDPM ancestral step (models/__init__.py:289-330):
 *   x0 = k1*(x - k2*e); clip; x = c1*x0 + c2*x (+ sigma*noise when noise != NULL). */
int evc_ddpm_step_f32(float* x, const float* e, const float* noise, long long n, float k1, float k2, float c1,
                      float c2, float sigma, int clip, void* stream);
/* DDIM step (models/__init__.py:163-166): x = c1*clip(k1*(x - k2*e)) + c2*e. */
int evc_ddim_step_f32(float* x, const float* e, long long n, float k1, float k2, float c1, float c2, int clip,
                      void* stream);
/* y = x + alpha*e  (final denoise call, models/__init__.py:333-335, with alpha = -sqrt(1-a)). */
int evc_axpy_f32(const float* x, const float* e, float* y, long long n, float alpha, void* stream);
/* PNDM transfer (models/pndm.py:19-33): y = clip(x + d*(cx*x - ce*e)). */
int evc_pndm_transfer_f32(const float* x, const float* e, float* y, long long n, float d, float cx, float ce,
                          int clip, void* stream);
/* y = w0*e0 + w1*e1 + w2*e2 + w3*e3 (Runge-Kutta / Adams-Bashforth combination, pndm.py:15,47). */
int evc_lincomb4_f32(const float* e0, const float* e1, const float* e2, const float* e3, float* y, long long n,
                     float w0, float w1, float w2, float w3, void* stream);
/* data_transform / inverse_data_transform (city_sender.py:232-244, function.py:73-82):
 * y = x*mul + add, optionally clamped to [lo, hi]. */
int evc_scale_clamp_f32(const float* x, float* y, long long n, float mul, float add, int clamp, float lo, float hi,
                        void* stream);

/* ---- ELIC helpers ---------------------------------------------------------------------------- */
/* out = a * sigmoid(b) + x (AttentionBlock.forward, ELICUtilis/layers/layers.py:247-252). */
int evc_gate_residual_f32(const float* a, const float* b, const float* x, float* out, long long n, void* stream);
/* GaussianConditional.build_indexes on the checkerboard half of a slice (Network.py:488-496):
 * for the anchor (parity 0) or non-anchor (parity 1) sites of scales [B][H][W][ld] at channel
 * offset c0..c0+C, write idx[b][c][h][w/2] = #table entries logic of compressai build_indexes and the
 * matching means[b][c][h][w/2]. */
int evc_elic_gather_params_f32(const float* ms, int ld, int mean_off, int scale_off, int C, int B, int H, int W,
                               int parity, const float* scale_table, int n_scales, int* idx, float* means,
                               void* stream);
/* y_hat[b][h][w][c0 + c] = symbols[b][c][h][w/2] + means (checkerboard sites of `parity`); other sites
 * untouched (Network.py:498-499, 523-524). */
int evc_elic_scatter_symbols_f32(const int* symbols, const float* means, float* y_hat, int ld, int c0, int C,
                                 int B, int H, int W, int parity, void* stream);
/* Encoder side (Network.py:399, 423 via compressai quantize "symbols"): symbols[b][c][h][w/2] =
 * round_half_even(y[b][h][w][c0 + c] - means[b][c][h][w/2]) at the checkerboard sites of `parity`. */
int evc_elic_quantize_f32(const float* y, int ld, int c0, const float* means, int C, int B, int H, int W,
                          int parity, int* symbols, void* stream);

/* ---- ELIC stride-2 convolutions, polyphase form (csrc/stride2.hip) ----------------------------------------------
 * compressai deconv() = ConvTranspose2d(k 5, stride 2, padding 2, output_padding 1) and conv() = Conv2d(k 5, stride 2,
 * padding 2) (reference Network.py:88-138).  Exactly the same sums as the direct forms, as ONE 3x3 stride-1 convolution
 * (evc_conv2d_nhwc_f32) on the low-resolution side plus a layout pass: 36 MACs per low-resolution pixel and channel pair
 * instead of the 100 of zero-insertion / decimation around a 5x5 "same" convolution.
 *   evc_deconv5x5s2_phase_weights_f32: ConvTranspose2d weight wt [Ci][Co][5][5] -> wp [4*Cp][CiPad][3][3] (phase-major
 *       output channels, zero taps where a phase has two; Cp >= Co, CiPad >= Ci), to be packed by evc_conv_pack_weights.
 *   evc_conv5x5s2_phase_weights_f32: Conv2d weight w [Co][Ci][5][5] -> wq [Co][4*Cq][3][3] over the space-to-depth input.
 *   evc_depth_to_space2_f32 / evc_space_to_depth2_f32: the layout passes (pure permutations, zero padding channels).
 *   evc_deconv5x5s2_f32: x [B][H][W][Ci] -> out [B][2H][2W][Co]; bias4 [4*Cp] = the bias repeated per phase.
 *   evc_conv5x5s2_f32:   x [B][2Ho][2Wo][ld_in] (first Ci channels) -> out [B][Ho][Wo][Co]. */
int evc_deconv5x5s2_phase_weights_f32(const float* wt, float* wp, int Ci, int Co, int Cp, int CiPad, void* stream);
int evc_conv5x5s2_phase_weights_f32(const float* w, float* wq, int Co, int Ci, int Cq, void* stream);
int evc_depth_to_space2_f32(const float* in, int ld_in, int Cp, float* out, int C, int B, int H, int W, void* stream);
int evc_space_to_depth2_f32(const float* in, int ld_in, int C, float* out, int ld_out, int Cq, int B, int H, int W,
                            void* stream);
long long evc_deconv5x5s2_workspace_bytes(int B, int H, int W, int Cp);
int evc_deconv5x5s2_f32(const float* x, const void* w_packed, int arith, const float* bias4, float* out, float* ws, int B,
                        int H, int W, int Ci, int Co, int act_out, void* stream);
long long evc_conv5x5s2_workspace_bytes(int B, int Ho, int Wo, int Ci);
int evc_conv5x5s2_f32(const float* x, int ld_in, const void* w_packed, int arith, const float* bias, float* out, float* ws,
                      int B, int Ho, int Wo, int Ci, int Co, int act_out, void* stream);

/* ---- LPIPS (AlexNet, v0.1): the perceptual distance of the sender's decision rule ------------------------------
 * Replaces lpips.LPIPS(net='alex') as built by city_sender.py:302 and called by decide_5to5_lpips (:376-406); the
 * algorithm is the one vendored in models/networks_basic.py:62-93 (PNetLin.forward, ScalingLayer), models/eval_models.py:35-37
 * (normalize_tensor) and models/pretrained_networks.py:56-94 (AlexNet slices), restated in oracle/lpips.py.  The five convolutions run on evc_conv2d_nhwc_f32; these
 * are the other pieces (csrc/lpips.hip):
 *   evc_im2col_nchw_f32   x (N, C, H, W) NCHW -> out (N, Ho, Wo, ld_out) rows of KH*KW patches in (c, ky, kx) order, the
 *                         columns C*KH*KW .. ld_out-1 zero; optional per-channel (x - shift[c]) / scale[c] (the ScalingLayer)
 *                         applied before the zero padding, as the convolution that follows would see it.  Ho = (H + 2 pad -
 *                         KH) / stride + 1.  Turns the stride-4 11x11 first convolution into a 1x1 convolution.
 *   evc_maxpool3s2_nhwc_f32   MaxPool2d(3, stride 2), no padding, floor: (N, H, W, C) -> (N, (H-3)/2+1, (W-3)/2+1, C), C % 4 == 0.
 *   evc_lpips_layer_f32   one feature tap: dist[n] (+)= mean over pixels of sum_c lin_w[c] * (f0/(|f0|_2 + 1e-10) -
 *                         f1/(|f1|_2 + 1e-10))^2, norms over the channels of a pixel; f0, f1: (N, HW, C) NHWC. */
int evc_im2col_nchw_f32(const float* x, float* out, int N, int C, int H, int W, int KH, int KW, int stride, int pad, int ld_out,
                        const float* shift, const float* scale, void* stream);
int evc_maxpool3s2_nhwc_f32(const float* x, float* out, int N, int H, int W, int C, void* stream);
int evc_lpips_layer_f32(const float* f0, const float* f1, const float* lin_w, float* dist, int N, int HW, int C, int accumulate,
                        void* stream);

/* ---- GDN (not on the decode path: g_s / g_a contain none) ---------------------------------
 * y = x * rsqrt(beta + gamma . x^2) (inverse: * sqrt) -- GDN.forward, ELICUtilis/layers/gdn.py:62-77; simplified != 0:
 * y = x / (beta + gamma . |x|) -- GDN1.forward, :95-106.  x, out: NHWC with C % 16 == 0; gamma_packed = the
 * re-parametrised (C, C) gamma packed as a 1x1 convolution weight by evc_conv_pack_weights (EVC_ARITH_F32 or _BF16X6);
 * beta: the re-parametrised (C,) vector; ws: evc_gdn_workspace_bytes() bytes. */
long long evc_gdn_workspace_bytes(int B, int H, int W, int C);
int evc_gdn_f32(const float* x, const void* gamma_packed, int arith, const float* beta, float* out, float* ws, int B,
                int H, int W, int C, int inverse, int simplified, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EVC_HIP_H */
