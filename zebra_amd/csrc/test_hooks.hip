// Test hooks: direct access to device primitives whose rare paths need their
// own tests (exact top-k selection under ties).  Not part of the product API.
#include "numba_sort.hpp"
#include "tppr_state.hpp"
#include "test_hooks.h"

using namespace zt;

namespace {

struct HookLds {
    double a[1536];
    int sel[64];
    int perm[1536];
    int stk[96];
    SortLds sort;
};

// one wavefront per case; mode 0 = production dispatch (fast / wave / seq),
// 1 = force the wave-parallel exact sort (LDS), 2 = force the sequential replay,
// 3 = force the register-resident replay (n <= 64), 4 = the replay on ranks, 5 / 6 = topk_reg, candidates adjacent / split over the two wave halves (n <= 63, k <= 31)
__global__ __launch_bounds__(64) void k_test_topk(const double *__restrict__ vals, int n, int k, int cases, int mode,
                                                  int *sel_out, int *path_out)
{
    __shared__ HookLds L;
    const int c = blockIdx.x;
    if (c >= cases) return;
    const int lane = lane_id();
    for (int q = lane; q < n; q += WAVE) L.a[q] = vals[(size_t)c * n + q];
    wave_sync();
    int path;
    const long long t0 = (long long)wall_clock64();
    if (mode == 0) {
        path = topk_select_wave(L.a, n, k, L.sel, L.sort, L.perm, L.stk);
    } else if (mode == 4) {                       // the rank-domain replay on its own, ties or not
        const double v = lane < n ? L.a[lane] : 0.0;
        int lt;
        (void)topk_rank_reg(v, n, k, L.sel, &lt);
        topk_ties_reg(lt, n, k, L.sel, L.sort);
        path = 4;
    } else if (mode == 5 || mode == 6) {          // the register-resident selection of merge_pair_reg:
        // 5 = candidates in lanes [0, n); 6 = split like the merge: the first half in lanes [0, n1), the rest from lane 32
        const int n1 = mode == 5 ? n : (n + 1) / 2;
        const int pos = lane < n1 ? lane : (mode == 6 && lane >= 32 && lane - 32 < n - n1 ? n1 + lane - 32 : -1);
        const u64 live = __ballot(pos >= 0);
        const double v = pos >= 0 ? L.a[pos] : 0.0;
        int slot;
        path = topk_reg(v, live, pos, n, k, L.sort, &slot);
        if (slot >= 0) L.sel[slot] = pos;
        wave_sync();
    } else if (mode == 3) {
        numba_argsort_reg(L.a, n, L.sort);
        if (lane < k) L.sel[lane] = L.sort.r2[n - k + lane];
        wave_sync();
        path = 3;
    } else if (mode == 1) {
        numba_argsort_wave(L.a, n, L.sort);
        if (lane < k) L.sel[lane] = L.sort.r2[n - k + lane];
        wave_sync();
        path = 1;
    } else {
        if (lane == 0) {
            numba_argsort_seq(L.a, n, L.perm, L.stk);
            for (int q = 0; q < k; ++q) L.sel[q] = L.perm[n - k + q];
        }
        wave_sync();
        path = 2;
    }
    const long long t1 = (long long)wall_clock64();
    if (lane < k) sel_out[(size_t)c * k + lane] = L.sel[lane];
    if (lane == 0) path_out[c] = path | ((int)(t1 - t0) << 8);   // bits 8..: duration in 10 ns ticks
}

}  // namespace

extern "C" int zt_test_topk(const double *vals_dev, int32_t n, int32_t k, int32_t cases, int32_t mode,
                            int32_t *sel_out_dev, int32_t *path_out_dev, void *stream)
{
    if (!vals_dev || !sel_out_dev || !path_out_dev || n < 2 || k < 1 || k >= n || k > 64 || cases < 1 || n > 1536 ||
        (mode == 1 && n > 128) || ((mode == 3 || mode == 4) && n > 64) || ((mode == 5 || mode == 6) && (n > 63 || k > 31))) {
        set_error("zt_test_topk: bad argument");
        return ZT_ERR_ARG;
    }
    k_test_topk<<<cases, 64, 0, (hipStream_t)stream>>>(vals_dev, n, k, cases, mode, sel_out_dev, path_out_dev);
    ZT_LAUNCH_CHECK();
    return ZT_OK;
}

// move the launch epoch of a T-PPR handle (e.g. next to its wrap-around: all tags are cleared when it reaches 2^17 - 1)
extern "C" int zt_test_set_epoch(zt_tppr *h, uint32_t epoch)
{
    if (!h) return ZT_ERR_ARG;
    h->epoch = epoch > EPOCH_MAX ? EPOCH_MAX : epoch;
    return ZT_OK;
}

// the dependency plan zt_tppr_plan made last (tppr_prepass.hip), copied to host arrays: wo / pflag / hv [3 * B] (B = the
// planned launch's edges), owner_of [B], chain_node / chain_len [16], chain_edges [16][2048]; *n_chains = the chains picked
extern "C" int zt_test_tppr_plan_dump(zt_tppr *h, int32_t *wo, int32_t *pflag, int32_t *hv, int32_t *owner_of,
                                      int32_t *chain_node, int32_t *chain_len, int32_t *chain_edges, int32_t *n_chains)
{
    if (!h) return ZT_ERR_ARG;
    const zt_tppr::PlanSet &P = h->set[h->next_set ^ 1];
    if (!P.valid) return ZT_ERR_ARG;
    ZT_HIP(hipDeviceSynchronize());
    const size_t a = (size_t)P.B * P.n_roles * sizeof(int);
    ZT_HIP(hipMemcpy(wo, P.wo, a, hipMemcpyDeviceToHost));
    ZT_HIP(hipMemcpy(pflag, P.pflag, a, hipMemcpyDeviceToHost));
    ZT_HIP(hipMemcpy(hv, P.hv, a, hipMemcpyDeviceToHost));
    ZT_HIP(hipMemcpy(owner_of, P.owner_of, (size_t)P.B * sizeof(int), hipMemcpyDeviceToHost));
    ZT_HIP(hipMemcpy(chain_node, P.chain_node, MAX_CHAINS * sizeof(int), hipMemcpyDeviceToHost));
    ZT_HIP(hipMemcpy(chain_len, P.chain_len, MAX_CHAINS * sizeof(int), hipMemcpyDeviceToHost));
    ZT_HIP(hipMemcpy(chain_edges, P.chain_edges, (size_t)MAX_CHAINS * CH_MAX * sizeof(int), hipMemcpyDeviceToHost));
    ZT_HIP(hipMemcpy(n_chains, P.ctl + 4, sizeof(int), hipMemcpyDeviceToHost));
    return ZT_OK;
}
