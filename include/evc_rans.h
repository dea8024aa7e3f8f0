/* evc_rans.h -- C ABI of libevc_rans.so: host-side range-ANS entropy coder for the ELIC key frames.
 *
 * Replaces the third-party C++ coder the reference reaches through compressai==1.1.5
 * (requirements.txt:17; not vendored): `RansEncoder.encode_with_indexes` / `RansDecoder.decode_with_indexes`
 * called from EntropyModel.compress / decompress at reference call sites Network.py:346-347, 400-401,
 * 424-428 (encode) and Network.py:450, 493-496, 514-517 (decode).
 * Format (restated from the published compressai / ryg_rans algorithm, parity unpinned -- see DESIGN.md):
 * rANS with a 64-bit state, 32-bit renormalisation words, 16-bit CDF precision; symbol value
 * v = symbol - offset[index]; values outside [0, cdf_size-2) are coded as the sentinel cdf_size-2
 * followed by a 4-bit-chunk bypass code; symbols are pushed in reverse so decoding runs forward.
 *
 * All pointers are HOST pointers.  cdfs is a row-major [n_cdfs][cdf_ld] int32 table.
 */
#ifndef EVC_RANS_H
#define EVC_RANS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define EVC_RANS_EINVAL (-1)    /* bad argument / index out of range */
#define EVC_RANS_ENOSPC (-2)    /* output buffer too small */
#define EVC_RANS_ECORRUPT (-3)  /* bitstream ends early or decodes outside its table */

const char* evc_rans_version(void);

/* Upper bound on the encoded size of n symbols (bytes). */
long long evc_rans_max_encoded_bytes(long long n);

/* Returns the number of bytes written to `out` (>= 8), or a negative error. */
long long evc_rans_encode_with_indexes(const int32_t* symbols, const int32_t* indexes, long long n,
                                       const int32_t* cdfs, int cdf_ld, const int32_t* cdf_sizes,
                                       const int32_t* offsets, int n_cdfs, uint8_t* out, long long out_cap);

/* Decodes n symbols into `symbols`; returns 0 or a negative error. */
int evc_rans_decode_with_indexes(const uint8_t* in, long long n_bytes, const int32_t* indexes, long long n,
                                 const int32_t* cdfs, int cdf_ld, const int32_t* cdf_sizes,
                                 const int32_t* offsets, int n_cdfs, int32_t* symbols);

/* pmf (float, length n) -> quantised CDF (int32, length n + 1, last = 1 << precision), the table
 * builder of compressai `pmf_to_quantized_cdf` (used when synthesising checkpoints; real ELIC
 * checkpoints ship their tables). Returns 0 or a negative error. */
int evc_pmf_to_quantized_cdf(const float* pmf, int n, int precision, int32_t* cdf);

#ifdef __cplusplus
}
#endif
#endif
