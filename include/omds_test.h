/*
 * omds_test.h -- test hooks of the MI355X-native MPPI rollout path.  NOT part of the product ABI: libomds_hip.so does not export
 * them.  They exist in libomds_hip_test.so (`make test-lib`: the same objects except capi, tail_kernel and train, which are compiled
 * with -DOMDS_TEST_HOOKS), which tests/ load explicitly (optimalmodulationds_amd._lib.load_test_hooks()).  Neither library
 * reads experiment environment variables; those exist only in `make experiment` builds (csrc/omds_internal.h).
 */
#ifndef OMDS_TEST_H
#define OMDS_TEST_H

#include "omds.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Damages what the screening network sees so that the run-time checks of the screened step (omds.h, "Screening of pass 1") have
 * something to catch.  what = 0: zeroes fragment `index` (1 KiB) of the fp16 weight pack; what = 1: shifts obstacle `index` by
 * `value` along x in the screening kernel's input table only (until the next omds_set_obstacles).  The fp32 kernels are never
 * touched.                                                                                                               */
OMDS_API int omds_screen_debug_corrupt(omds_ctx* ctx, int what, int index, float value);
/* Process-wide: the tile shape of the step's tail kernels instead of the launcher's own choice -- tail_sel_rows in {0, 4, 16, 32}
 * for the screened step (4 = the backward on 4-row groups; ReLU networks without skip concatenations), tail_rows in {0, 4, 16, 32}
 * for the unscreened one (4 = forward and backward on 4-row groups); 0 = the launcher chooses again.  Every shape computes the same bits per row: the tests run them against
 * each other.                                                                                                             */
OMDS_API int omds_debug_force_tile_rows(int tail_sel_rows, int tail_rows);
/* Process-wide: every product of the trainer (omds_trainer_step / _eval) on the general GEMM kernel instead of the special-shape
 * kernels (tall 256-wide layers, thin first / last layers, thin weight gradients).  All of them sum in ascending k from zero, so
 * the runs must agree bit for bit: tests/test_gpu_train.py compares them on a batch that is not a multiple of the tile height. */
OMDS_API int omds_debug_trainer_general_gemm(int on);
/* The HOST half of omds_set_mlp_ex alone: argument validation, zero-padding to the kernels' width and every MFMA fragment pack
 * (fp32 forward / backward, 16-row, 4-row-group, fp16 screening slices), with no device and no context -- the sanitizer build
 * (`make asan`) runs it on the CPU (tests/test_asan_cpu.py).  *checksum = FNV-1a over all packs, *bytes = their total size (NULL =
 * skip); the message of a failure through omds_last_error(NULL).                                                          */
OMDS_API int omds_test_pack_mlp(int n_dof, int n_linear, const int32_t* in_dims, const int32_t* out_dims, const float* const* W,
                                const float* const* b, int act, float out_div, int n_skips, const int32_t* skip_after,
                                uint64_t* checksum, int64_t* bytes);

#ifdef __cplusplus
}
#endif
#endif /* OMDS_TEST_H */
