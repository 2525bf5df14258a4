/*
 * pfac_module.h -- the plugin seam between libpfac.so (host API, pattern
 * compiler, tables) and the per-architecture kernel module libpfac_gfx950.so.
 *
 * The reference binds the same four C symbols out of libpfac_sm_<NN>.so with
 * dlopen/dlsym at PFAC_create() time (PFAC/include/PFAC_P.h:41-45,
 * PFAC/src/PFAC.cpp:158-201); the names (including the historical spelling
 * "warpper") and argument lists are kept so the seam is recognisable and a
 * module built for another architecture could be dropped in beside this one.
 * The module file name is derived from hipDeviceProp_t::gcnArchName
 * ("gfx950:sramecc+:xnack-" -> "libpfac_gfx950.so") instead of 10*major+minor.
 *
 * The texture-bind mutex the reference module calls back into
 * (PFAC_P.h:217-226) has no gfx950 counterpart: a buffer-resource descriptor
 * is built in SGPRs per launch, there is no process-global binding to guard.
 */
#ifndef PFAC_MODULE_H_
#define PFAC_MODULE_H_

#include "PFAC.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ref PFAC_P.h:41-42 (PFAC_kernel_protoType): full result, one int per input byte. */
typedef PFAC_status_t (*PFAC_kernel_protoType)(PFAC_handle_t handle, char *d_input_string,
                                               size_t input_size, int *d_matched_result);

/* ref PFAC_P.h:44-45 (PFAC_reduce_kernel_protoType): compacted (id, position) output. */
typedef PFAC_status_t (*PFAC_reduce_kernel_protoType)(PFAC_handle_t handle, int *d_input_string,
                                                      int input_size, int *d_match_result,
                                                      int *d_pos, int *h_num_matched,
                                                      int *h_match_result, int *h_pos);

/* Exported by libpfac_gfx950.so.
 * ref PFAC_kernel.cu:90-244 (dense table) and PFAC_kernel_spaceDriven.cu:149-348 (hashed). */
PFAC_status_t PFAC_kernel_timeDriven_warpper(PFAC_handle_t handle, char *d_input_string,
                                             size_t input_size, int *d_matched_result);
PFAC_status_t PFAC_kernel_spaceDriven_warpper(PFAC_handle_t handle, char *d_input_string,
                                              size_t input_size, int *d_matched_result);

/* ref PFAC_reduce_kernel.cu:172-295 and PFAC_reduce_inplace_kernel.cu:155-323. */
PFAC_status_t PFAC_reduce_kernel(PFAC_handle_t handle, int *d_input_string, int input_size,
                                 int *d_match_result, int *d_pos, int *h_num_matched,
                                 int *h_match_result, int *h_pos);
PFAC_status_t PFAC_reduce_inplace_kernel(PFAC_handle_t handle, int *d_input_string, int input_size,
                                         int *d_match_result, int *d_pos, int *h_num_matched,
                                         int *h_match_result, int *h_pos);

/* Measurement only (no reference counterpart): the traffic shape of the match path with nothing else in it -- every
 * wave reads 1 KiB of d_in and writes 4 KiB of zeros to d_out, non-temporal.  Returns the average milliseconds of
 * `launches` launches over the first n bytes (a multiple of 4096) of d_in, or a negative value on a HIP error.
 * bench.py reports it next to the scan as "what this part sustains for 1 B read : 4 B written" (SURVEY 8d). */
double PFACX_streamProbe(const void *d_in, void *d_out, size_t n, int launches);

/* Measurement only: the compile-time shape of this module (block size, writer waves, walk sets, queue and list capacity,
 * front geometry, ablation / timing builds) as one static string for the bench record. */
const char *PFACX_buildInfo(void);

#ifdef __cplusplus
}
#endif

#endif /* PFAC_MODULE_H_ */
