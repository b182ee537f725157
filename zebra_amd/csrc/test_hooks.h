/* Test hooks: libzebra_amd_testhooks.so (zebra_amd/csrc/test_hooks.hip), built beside the product library and loaded
 * by the tests only.  Nothing here is part of the product ABI (include/zebra_amd.h). */
#ifndef ZEBRA_AMD_TEST_HOOKS_H
#define ZEBRA_AMD_TEST_HOOKS_H
#include "../../include/zebra_amd.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------ */
/* Test hook (not product API): the exact top-k selection primitive on its    */
/* own.  vals [cases][n] float64; writes np.argsort(vals[c])[-k:] under       */
/* numba's quicksort semantics into sel_out [cases][k] and the path taken     */
/* (0 fast rank, 1 wave-parallel exact in LDS, 2 sequential exact, 3 exact in  */
/* registers on values, 4 exact in registers on ranks) into path_out.          */
/* mode: 0 = production dispatch, 1/2/3/4 = force that path, 5 / 6 = the        */
/* register-resident selection of the merge (one candidate per lane, adjacent  */
/* lanes / split over the two wave halves; n <= 63, k <= 31).                  */
/* ------------------------------------------------------------------------ */
int zt_test_topk(const double *vals_dev, int32_t n, int32_t k, int32_t cases,
                 int32_t mode, int32_t *sel_out_dev, int32_t *path_out_dev,
                 void *stream);
/* Test hook: set the T-PPR handle's launch epoch (the row tags' high bits), to
 * exercise the wrap-around (all tags are cleared when it reaches 2^18 - 1). */
int zt_test_set_epoch(zt_tppr *h, uint32_t epoch);
/* Test hook: the dependency plan zt_tppr_plan made last, copied to host arrays -- wo / pflag / hv [n_roles * B] (B = the
 * planned launch's edges), owner_of [B], chain_node / chain_len [16], chain_edges [16][2048], *n_chains. */
int zt_test_tppr_plan_dump(zt_tppr *h, int32_t *wo, int32_t *pflag, int32_t *hv, int32_t *owner_of, int32_t *chain_node,
                           int32_t *chain_len, int32_t *chain_edges, int32_t *n_chains);

#ifdef __cplusplus
}
#endif
#endif
