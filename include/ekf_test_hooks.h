/*
 * ekf_test_hooks.h -- fault injection for the test suite.  NOT part of the drop-in boundary (include/ekf_engine.h): nothing a host
 * of the reference would call.  The symbols are exported by libekf_engine.so so that the tests can reach them through the same
 * C ABI as everything else.
 */
#ifndef EKF_TEST_HOOKS_H
#define EKF_TEST_HOOKS_H

#include "ekf_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The NEXT persistent Cholesky sweep of this engine (csrc/chol_persist.h) runs without its chain workgroup -- the situation of a
 * role that never became resident.  Every other role then waits for a hand-off that cannot come; the bounded waits (30 ms) must end
 * the kernel with the sticky code EKF_ERR_TIMEOUT, the kernels behind the sweep must leave x and P untouched, and the engine must
 * run the update again on the launch-per-panel sweep: the caller sees EKF_OK, the oracle's result, and one more
 * ekf_get_sweep_retries.  One sweep only.  No counterpart in the reference. */
int ekf_debug_stall_next_sweep(EkfEngine *e);
/* ... the same for the persistent sweep AFTER the next `skip` ones (skip = 1: the second update of the next EKF::step). */
int ekf_debug_stall_sweep_after(EkfEngine *e, int skip);
/* Exact configurations: how many 16-row x 32-column pieces of digit plane 0 of B held anything but zeros in the LAST update, of how
 * many (the downdate skips the digit products of the all-zero ones: csrc/kernels_pexact.hip, px_flag_plane0). */
int ekf_debug_plane0_pieces(EkfEngine *e, int *nonzero, int *total);
/* on != 0: the exact downdate and the int8 GEMM B = inv(L) G multiply EVERY digit product, the tables of zero pieces are not
 * consulted (they are still written).  The results must be the same bit for bit either way -- which is what the suite checks with
 * two engines on the same frames (tests/test_gpu_exact_edge_cases.py). */
int ekf_debug_dense_products(EkfEngine *e, int on);

#ifdef __cplusplus
}
#endif
#endif /* EKF_TEST_HOOKS_H */
