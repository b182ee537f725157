// The dependency prepass of a big launch as ONE cooperative kernel (round 5): built, bit-exact, NOT faster than the eleven
// launches -- with fences it costs the aggregation beside it, without them it is slower itself --
// and therefore not part of the product library: compiled only into variant builds
//     tools/build_variant.sh prepasscoop "-DZT_PREPASS_COOP_VARIANT -I/root/repo/tools/exp/variants" tppr_prepass.hip
// (tppr_prepass.hip includes it inside its anonymous namespace).
#pragma once

// The whole prepass of a BIG launch as one kernel of COOP_WGS workgroups (round 5): the same steps on the same arrays, separated
// by grid barriers -- a counter in the set's control words that only ever grows (the host knows its value before the launch) --
// instead of by ten kernel boundaries: a C5 batch's prepass took 110 us as eleven launches, most of it launch latency and the
// gaps between them, and the first batch of a timed region waits for it.  All workgroups must be resident for a barrier to
// open: COOP_WGS x COOP_THREADS is a small grid (it fits beside anything but a kernel that owns every register of every CU, and
// then it waits for that kernel's end like the eleven launches did); waiting workgroups sleep between polls.
constexpr int COOP_WGS = 24, COOP_THREADS = 512, COOP_BARRIERS = 8;
constexpr int CTL_BAR = 11;
__device__ __forceinline__ void coop_barrier(int *bar, unsigned target)
{
    // no fence: what the steps hand each other travels as write-through stores and sc1 loads (pre_st / pre_ld <true>) or as
    // device-scope atomics; a thread's stores have been acknowledged before it arrives
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(bar, 1);
        while ((int)((unsigned)ld_agent(bar) - target) < 0) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
}

__global__ __launch_bounds__(COOP_THREADS) void k_prepass_coop(
    const int *__restrict__ nodes, const long long *__restrict__ eidx, long long role_stride, int B, int n_roles,
    long long N, int *cnt, int *slot, int *off, int *list, int *wo, int *pflag, int *nxt, int *ctl, int *latch,
    int *hot_node, int *hot_cnt, int *chain_of, int *chain_node, int *chain_len, int *chain_edges, int *owner_of,
    int *hv, int max_chains, int big_min, int grid, int n_models, unsigned bar_base)
{
    __shared__ int sort_s[DEPS_SORT_MAX], sort_t[2 * COOP_THREADS];
    const int tid = threadIdx.x, A = B * n_roles, G = gridDim.x;
    const int gtid = blockIdx.x * COOP_THREADS + tid, gthr = G * COOP_THREADS;
    int *bar = ctl + CTL_BAR;
    unsigned target = bar_base;
    if (gtid == 0) { st_agent(ctl + 2, 0); st_agent(ctl + 13, 0); }                                  // k_plan_begin
    coop_barrier(bar, target += G);
    for (int a = gtid; a < A; a += gthr) d_count<true>(a, nodes, eidx, role_stride, B, n_roles, N, cnt, slot, ctl, latch);
    coop_barrier(bar, target += G);
    for (int a = gtid; a < A; a += gthr) d_reserve<true>(a, nodes, role_stride, B, n_roles, cnt, off, slot, ctl, hot_node, hot_cnt, big_min);
    coop_barrier(bar, target += G);
    for (int a = gtid; a < A; a += gthr) d_fill<true>(a, nodes, role_stride, B, n_roles, off, slot, list);
    coop_barrier(bar, target += G);
    {
        const int n_big = ld_agent(&ctl[5]);
        // the big groups first, one workgroup each (the sort is the longest single piece of the prepass), then the per-access
        // walk of everybody else; the last workgroup picks the chains meanwhile (it needs the candidates only)
        if (n_big <= MAX_BIG)
            for (int g = blockIdx.x; g < n_big; g += G)
                d_deps_group<true>(tid, COOP_THREADS, ld_agent(hot_node + MAX_HOT + g), B, cnt, off, list, wo, pflag, nxt, sort_s, sort_t);
        for (int a = gtid; a < A; a += gthr) d_deps<true>(a, nodes, role_stride, B, n_roles, cnt, off, slot, list, wo, pflag, nxt, n_big, big_min);
        if ((int)blockIdx.x == G - 1) d_hot_select<true>(tid, ctl, hot_node, hot_cnt, chain_of, chain_node, chain_len, max_chains, grid, B * n_models, n_models);
    }
    coop_barrier(bar, target += G);
    for (int i = gtid; i < B; i += gthr) d_own<true>(i, nodes, role_stride, B, cnt, slot, wo, chain_of, chain_len, chain_edges, owner_of);
    coop_barrier(bar, target += G);
    for (int a = gtid; a < A; a += gthr) d_hubacc<true>(a, nodes, role_stride, B, n_roles, slot, wo, chain_of, chain_len, hv);
    coop_barrier(bar, target += G);
    for (int a = gtid; a < (A > MAX_HOT ? A : MAX_HOT); a += gthr)
        d_cleanup<true>(a, nodes, role_stride, B, n_roles, slot, cnt, ctl, hot_node, chain_of);
    // the last workgroup through resets the control words (k_reset_ctl: after every cleanup thread has read ctl[3])
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        target += G;
        if ((unsigned)atomicAdd(bar, 1) + 1u == target) {
            for (int q = 0; q < 6; ++q) if (q != 2 && q != 4) st_agent(ctl + q, 0);
        }
    }
}

