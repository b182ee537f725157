// The dependency prepass of a streaming T-PPR launch (reference utils/util.py:495-574 applies a batch's edges one by
// one; here the order is made explicit before the update kernel runs): groups the batch's node accesses by node, gives
// every access the tag of the last earlier writer of its node, picks the hub chains.  Reads node and edge ids only, so it
// runs ahead of the update on another stream (zt_tppr_plan).
#include "tppr_state.hpp"

using namespace zt;

namespace {

// ---------------------------------------------------------------- prepass ----
// One access = (role r, edge i) of the chunk; a = r*B + i.  Roles 0/1 (source,
// destination) write their node's row, role 2 (negative) only reads it.  An
// edge's second access to the same node (self-loop, negative == endpoint) is a
// SHADOW: it is not entered into the node's group.
// K1: validate ids; count accesses per node; remember each access' slot.
// K0: a set is planned afresh: the status of the launch it last served is history (the handle's latch keeps it)
__global__ void k_plan_begin(int *ctl) { if (threadIdx.x == 0) { ctl[2] = 0; ctl[13] = 0; } }

// Accesses to the arrays one prepass step leaves for the next.  SC = false: plain (the steps are kernels of their own, or
// one workgroup).  SC = true (k_prepass_coop): relaxed agent-scope accesses -- write-through stores, loads that do not
// trust this compute unit's L1 or a remote XCD's copy -- so that a grid barrier needs no fence: a thread's stores have
// been acknowledged when it arrives (s_waitcnt vmcnt(0)), and what it loads afterwards comes from where they went.
template <bool SC> __device__ __forceinline__ int pre_ld(const int *p) { return SC ? ld_agent(p) : *p; }
template <bool SC> __device__ __forceinline__ void pre_st(int *p, int v) { if (SC) st_agent(p, v); else *p = v; }

template <bool SC = false>
__device__ __forceinline__ void d_count(int a, const int *__restrict__ nodes, const long long *__restrict__ eidx,
                                        long long role_stride, int B, int n_roles, long long N, int *cnt, int *slot,
                                        int *ctl, int *latch)
{
    if (a >= B * n_roles) return;
    const int r = a / B, i = a % B;
    const int x = nodes[(long long)r * role_stride + i];
    bool ok = x >= 0 && x < N;
    if (r == 0) {
        const long long e = eidx[i];
        ok = ok && e >= 0 && e <= 0x7fffffffll;
    }
    if (!ok) {
        atomicExch(&ctl[2], ZT_ERR_RANGE);
        latch_failure(latch, ZT_ERR_RANGE);
        pre_st<SC>(slot + a, -1);
        return;
    }
    bool shadow = false;
    if (r >= 1) shadow = nodes[i] == x;                                    // same as the source
    if (r == 2) shadow = shadow || nodes[role_stride + i] == x;           // same as the destination
    if (shadow) { pre_st<SC>(slot + a, -2); return; }
    pre_st<SC>(slot + a, atomicAdd(&cnt[x], 1));
}

__global__ void k_count(const int *__restrict__ nodes, const long long *__restrict__ eidx, long long role_stride,
                        int B, int n_roles, long long N, int *cnt, int *slot, int *ctl, int *latch)
{
    d_count(blockIdx.x * blockDim.x + threadIdx.x, nodes, eidx, role_stride, B, n_roles, N, cnt, slot, ctl, latch);
}

// K2: the first access of each node reserves a contiguous range of `list`.
template <bool SC = false>
__device__ __forceinline__ void d_reserve(int a, const int *__restrict__ nodes, long long role_stride, int B, int n_roles,
                                          const int *cnt, int *off, const int *slot, int *ctl, int *hot_node, int *hot_cnt,
                                          int big_min)
{
    if (a >= B * n_roles) return;
    if (pre_ld<SC>(slot + a) == 0) {
        const int x = nodes[(long long)(a / B) * role_stride + a % B];
        const int c = pre_ld<SC>(cnt + x);
        pre_st<SC>(off + x, atomicAdd(&ctl[0], c));
        if (c >= big_min && c <= DEPS_SORT_MAX) {  // a big group: its dependencies come from a sort (d_deps_group)
            const int bi = atomicAdd(&ctl[5], 1);
            if (bi < MAX_BIG) pre_st<SC>(hot_node + MAX_HOT + bi, x);
        }
        if (c >= HOT_MIN) {                        // hub candidate
            const int hi = atomicAdd(&ctl[3], 1);
            if (hi < MAX_HOT) { pre_st<SC>(hot_node + hi, x); pre_st<SC>(hot_cnt + hi, c); }
        }
    }
}

__global__ void k_reserve(const int *__restrict__ nodes, long long role_stride, int B, int n_roles, const int *cnt,
                          int *off, const int *slot, int *ctl, int *hot_node, int *hot_cnt, int big_min)
{
    d_reserve(blockIdx.x * blockDim.x + threadIdx.x, nodes, role_stride, B, n_roles, cnt, off, slot, ctl, hot_node, hot_cnt, big_min);
}

// How many hub chains a launch should run (round 4).  A chain workgroup is taken from the general queue, and the general
// queue has to get through the launch's `tasks` (edge, model) tasks -- ~16 us of a wave's time each -- in the time the longest
// chain needs (~1.5 us per hop, `top` hops): it wants tasks * 16 / (8 waves * 1.5 * top) = 4 tasks / (3 top) workgroups.
// What is left of the grid goes to chains (at least MIN_CHAINS of them).  C5 on 64 workgroups: 32 768 tasks against a hub of
// ~830 accesses per four-batch launch -> 53 general workgroups, 5 chains per model (measured: 5 .. 8 chains 0.353 ms/step,
// 12 0.382, 16 0.440; on 96 workgroups 16 chains fit: 0.352); C3: 4 800 tasks, ~480 -> 14 general workgroups, 16 chains.
constexpr int MIN_CHAINS = 4;
__device__ __forceinline__ int d_chain_budget(int top, int grid, int tasks, int n_models, int max_chains)
{
    if (top <= 0 || grid <= 0) return max_chains;
    const int general = (4 * tasks + 3 * top - 1) / (3 * top);
    int n = (grid - general) / (n_models > 0 ? n_models : 1);
    if (n < MIN_CHAINS) n = MIN_CHAINS;
    return n < max_chains ? n : max_chains;
}

// K2b (one wavefront pair): keep the MAX_CHAINS most-touched candidates as chains.
template <bool SC = false>
__device__ __forceinline__ void d_hot_select(int t, int *ctl, const int *hot_node, const int *hot_cnt, int *chain_of,
                                             int *chain_node, int *chain_len, int max_chains, int grid, int tasks, int n_models)
{
    if (t >= MAX_HOT) return;                      // MAX_HOT threads take part
    int nh = pre_ld<SC>(ctl + 3);
    nh = nh < MAX_HOT ? nh : MAX_HOT;
    {   // the chain budget of this launch (every thread works it out alike: at most MAX_HOT reads)
        int top = 0;
        for (int q = 0; q < nh; ++q) { const int cq = pre_ld<SC>(hot_cnt + q); top = cq > top ? cq : top; }
        max_chains = d_chain_budget(top, grid, tasks, n_models, max_chains);
    }
    if (t < MAX_CHAINS) pre_st<SC>(chain_len + t, 0);
    if (pre_ld<SC>(ctl + 2) == ZT_ERR_RANGE) { if (t == 0) pre_st<SC>(ctl + 4, 0); return; }
    int rank = 0;
    if (t < nh) {
        const int c = pre_ld<SC>(hot_cnt + t), x = pre_ld<SC>(hot_node + t);
        for (int q = 0; q < nh; ++q) {
            const int cq = pre_ld<SC>(hot_cnt + q);
            rank += (cq > c || (cq == c && pre_ld<SC>(hot_node + q) < x)) ? 1 : 0;
        }
        if (rank < max_chains) { pre_st<SC>(chain_of + x, rank); pre_st<SC>(chain_node + rank, x); }
    }
    if (t == 0) pre_st<SC>(ctl + 4, nh < max_chains ? nh : max_chains);
}



__global__ void k_hot_select(int *ctl, const int *hot_node, const int *hot_cnt, int *chain_of, int *chain_node,
                             int *chain_len, int max_chains, int grid, int tasks, int n_models)
{
    d_hot_select(threadIdx.x, ctl, hot_node, hot_cnt, chain_of, chain_node, chain_len, max_chains, grid, tasks, n_models);
}

// K2c: every edge with a hub endpoint enters that hub's chain at position = the hub's writer ordinal there (all earlier
// writers of a hub are its chain's edges, so the ordinal IS the position: no sort).  An edge between two hubs enters both
// chains -- each chain applies its own hub's update and takes the other hub's row from the other chain's versions; the
// chain of the more-touched hub is the edge's OWNER: its partner task in the general queue emits the edge's rows.
// A chain holds CH_MAX edges; a hub's later edges go through the general queue and the row in `rows`.
template <bool SC = false>
__device__ __forceinline__ void d_own(int i, const int *__restrict__ nodes, long long role_stride, int B, const int *cnt,
                                      const int *slot, const int *wo, const int *chain_of, int *chain_len, int *chain_edges,
                                      int *owner_of)
{
    if (i >= B) return;
    int owner = -1;
    if (pre_ld<SC>(slot + i) >= 0 && pre_ld<SC>(slot + B + i) != -1) {        // valid edge
        const int u = nodes[i], v = nodes[role_stride + i];
        const int cu = pre_ld<SC>(chain_of + u), cv = v != u ? pre_ld<SC>(chain_of + v) : -1;
        const int pu = pre_ld<SC>(wo + i), pv = pre_ld<SC>(wo + B + i);
        const bool in_u = cu >= 0 && pu < CH_MAX, in_v = cv >= 0 && pv < CH_MAX;
        if (in_u) { chain_edges[cu * CH_MAX + pu] = i; atomicMax(&chain_len[cu], pu + 1); }
        if (in_v) { chain_edges[cv * CH_MAX + pv] = i; atomicMax(&chain_len[cv], pv + 1); }
        if (in_u && (!in_v || pre_ld<SC>(cnt + u) >= pre_ld<SC>(cnt + v))) owner = cu;
        else if (in_v) owner = cv;
    }
    owner_of[i] = owner;
}

__global__ void k_own(const int *__restrict__ nodes, long long role_stride, int B, const int *cnt, const int *slot,
                      const int *wo, const int *chain_of, int *chain_len, int *chain_edges, int *owner_of)
{
    d_own(blockIdx.x * blockDim.x + threadIdx.x, nodes, role_stride, B, cnt, slot, wo, chain_of, chain_len, chain_edges, owner_of);
}

// K2d: per access, the chain that holds the node's row by version (zt_tppr::hv): a writer access that is a chain position,
// a reader access (negative sample) up to the chain's last version.  (After K2c: chain_len is final.)
template <bool SC = false>
__device__ __forceinline__ void d_hubacc(int a, const int *__restrict__ nodes, long long role_stride, int B, int n_roles,
                                         const int *slot, const int *wo, const int *chain_of, const int *chain_len, int *hv)
{
    if (a >= B * n_roles) return;
    int c = -1;
    if (pre_ld<SC>(slot + a) >= 0) {
        const int r = a / B;
        const int cc = pre_ld<SC>(chain_of + nodes[(long long)r * role_stride + a % B]);
        if (cc >= 0) {
            const int len = pre_ld<SC>(chain_len + cc), w = pre_ld<SC>(wo + a);
            if (len > 0 && (r < 2 ? w < CH_MAX : w <= len)) c = cc;
        }
    }
    hv[a] = c;
}

__global__ void k_hubacc(const int *__restrict__ nodes, long long role_stride, int B, int n_roles, const int *slot,
                         const int *wo, const int *chain_of, const int *chain_len, int *hv)
{
    d_hubacc(blockIdx.x * blockDim.x + threadIdx.x, nodes, role_stride, B, n_roles, slot, wo, chain_of, chain_len, hv);
}

// K3: scatter accesses into their node's range, encoded (edge << 2) | role.
template <bool SC = false>
__device__ __forceinline__ void d_fill(int a, const int *__restrict__ nodes, long long role_stride, int B, int n_roles,
                                       const int *off, const int *slot, int *list)
{
    if (a >= B * n_roles) return;
    const int s = pre_ld<SC>(slot + a);
    if (s < 0) return;
    pre_st<SC>(list + pre_ld<SC>(off + nodes[(long long)(a / B) * role_stride + a % B]) + s, ((a % B) << 2) | (a / B));   // (edge << 2) | role
}

__global__ void k_fill(const int *__restrict__ nodes, long long role_stride, int B, int n_roles, const int *off,
                       const int *slot, int *list)
{
    d_fill(blockIdx.x * blockDim.x + threadIdx.x, nodes, role_stride, B, n_roles, off, slot, list);
}

// K4: per access, from its node's group:
//   wo    = number of writer accesses by earlier edges  (= ordinal of the last earlier writer)
//   pflag = the latest earlier edge touching the node, if that access was a reader, else -1
//   nxt   = how many later edges touch the node (the length of the chain waiting for this access)
template <bool SC = false>
__device__ __forceinline__ void d_deps(int a, const int *__restrict__ nodes, long long role_stride, int B, int n_roles,
                                       const int *cnt, const int *off, const int *slot, const int *list, int *wo,
                                       int *pflag, int *nxt, int n_big, int big_min)
{
    if (a >= B * n_roles) return;
    if (pre_ld<SC>(slot + a) < 0) { pre_st<SC>(wo + a, 0); pre_st<SC>(pflag + a, -1); pre_st<SC>(nxt + a, 0); return; }
    const int x = nodes[(long long)(a / B) * role_stride + a % B];
    const int o = pre_ld<SC>(off + x), c = pre_ld<SC>(cnt + x);
    if (c >= big_min && c <= DEPS_SORT_MAX && n_big <= MAX_BIG) return;    // d_deps_group's (every big group is on the list)
    const int me = a % B;
    int best = -1, best_role = 0, writers = 0, nx = 0;
    // 16 list entries are fetched before any is used: a hub's group has hundreds of members and the
    // loop is otherwise one L2 round trip per entry
    for (int p0 = 0; p0 < c; p0 += 16) {
        int b[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) b[t] = pre_ld<SC>(list + o + ((p0 + t) < c ? (p0 + t) : (c - 1)));
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            if (p0 + t >= c) break;
            const int e = b[t] >> 2, r = b[t] & 3;
            if (e < me) {
                writers += (r < 2) ? 1 : 0;
                if (e > best) { best = e; best_role = r; }
            } else if (e > me) {
                ++nx;
            }
        }
    }
    pre_st<SC>(wo + a, writers);
    pre_st<SC>(nxt + a, nx);
    pre_st<SC>(pflag + a, (best >= 0 && best_role == 2) ? best : -1);
}

__global__ void k_deps(const int *__restrict__ nodes, long long role_stride, int B, int n_roles, const int *cnt,
                       const int *off, const int *slot, const int *list, int *wo, int *pflag, int *nxt, const int *ctl,
                       int big_min)
{
    d_deps(blockIdx.x * blockDim.x + threadIdx.x, nodes, role_stride, B, n_roles, cnt, off, slot, list, wo, pflag, nxt, ctl[5], big_min);
}

// K4b: the same three numbers for ALL members of one big group at once.  d_deps walks the whole group per access -- a hub
// with c accesses costs c^2 list reads, and the thread that owns one of them c / 16 dependent round trips: 70 us for
// the 400 members of C5's hub in a two-batch launch, 410 us in the single-workgroup form.  A group's members are
// (edge << 2 | role) words with distinct edges (an edge's second access to a node is a shadow and not in the group), so
// after a sort by value the member at position p has exactly p earlier edges: wo = writers among [0, p) (a prefix
// count), the latest earlier access is the member at p - 1, nxt = c - 1 - p.  One workgroup sorts the group in LDS
// (bitonic, padded to a power of two) and scans the writer flags.  `s` holds DEPS_SORT_MAX ints, `t` two per thread.
template <bool SC = false>
__device__ __forceinline__ void d_deps_group(int tid, int nthr, int x, int B, const int *cnt, const int *off, const int *list,
                                             int *wo, int *pflag, int *nxt, int *s, int *t)
{
    const int c = pre_ld<SC>(cnt + x), o = pre_ld<SC>(off + x);
    int P = 64;
    while (P < c) P <<= 1;
    for (int i = tid; i < P; i += nthr) s[i] = i < c ? pre_ld<SC>(list + o + i) : 0x7fffffff;
    __syncthreads();
    for (int k2 = 2; k2 <= P; k2 <<= 1)
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < P; i += nthr) {
                const int l = i ^ j;
                if (l > i) {
                    const int a = s[i], b = s[l];
                    if ((a > b) == ((i & k2) == 0)) { s[i] = b; s[l] = a; }
                }
            }
            __syncthreads();
        }
    // writers among [0, p): every thread counts a contiguous chunk, the chunk sums are scanned (Hillis-Steele between the
    // two halves of t: 2 * nthr ints)
    const int chunk = (P + nthr - 1) / nthr, b0 = tid * chunk;
    int loc = 0;
    for (int q = 0; q < chunk; ++q) loc += (b0 + q < c && (s[b0 + q] & 3) < 2) ? 1 : 0;
    int *src = t, *dst = t + nthr;
    src[tid] = loc;
    __syncthreads();
    for (int d = 1; d < nthr; d <<= 1) {
        dst[tid] = src[tid] + (tid >= d ? src[tid - d] : 0);
        __syncthreads();
        int *sw = src; src = dst; dst = sw;
    }
    int run = src[tid] - loc;
    for (int q = 0; q < chunk; ++q) {
        const int p = b0 + q;
        if (p >= c) break;
        const int v = s[p], e = v >> 2, r = v & 3, a = r * B + e;
        pre_st<SC>(wo + a, run);
        run += r < 2 ? 1 : 0;
        pre_st<SC>(nxt + a, c - 1 - p);
        pre_st<SC>(pflag + a, (p > 0 && (s[p - 1] & 3) == 2) ? (s[p - 1] >> 2) : -1);
    }
    __syncthreads();                                   // (the arrays are reused for the next group)
}

constexpr int DEPS_BIG_THREADS = 256;
__global__ __launch_bounds__(DEPS_BIG_THREADS) void k_deps_big(int B, const int *cnt, const int *off, const int *list, int *wo,
                                                               int *pflag, int *nxt, const int *ctl, const int *big_node)
{
    __shared__ int s[DEPS_SORT_MAX], t[2 * DEPS_BIG_THREADS];
    const int n_big = ctl[5];
    if (n_big > MAX_BIG) return;                       // the list overflowed: d_deps has taken every access
    for (int g = blockIdx.x; g < n_big; g += gridDim.x)
        d_deps_group(threadIdx.x, DEPS_BIG_THREADS, big_node[g], B, cnt, off, list, wo, pflag, nxt, s, t);
}

// K5: restore the per-node counters and the control words for the next call.
template <bool SC = false>
__device__ __forceinline__ void d_cleanup(int a, const int *__restrict__ nodes, long long role_stride, int B, int n_roles,
                                          const int *slot, int *cnt, int *ctl, const int *hot_node, int *chain_of)
{
    if (a < MAX_HOT && a < pre_ld<SC>(ctl + 3)) pre_st<SC>(chain_of + pre_ld<SC>(hot_node + a), -1);
    if (a >= B * n_roles) return;
    if (pre_ld<SC>(slot + a) == 0) pre_st<SC>(cnt + nodes[(long long)(a / B) * role_stride + a % B], 0);
}

__global__ void k_cleanup(const int *__restrict__ nodes, long long role_stride, int B, int n_roles, const int *slot,
                          int *cnt, int *ctl, const int *hot_node, int *chain_of)
{
    d_cleanup(blockIdx.x * blockDim.x + threadIdx.x, nodes, role_stride, B, n_roles, slot, cnt, ctl, hot_node, chain_of);
}

// K6: control words back to zero (after k_cleanup has read ctl[3]); ctl[4] = chains stays for k_stream.  ([5]: big groups)
__global__ void k_reset_ctl(int *ctl) { if (threadIdx.x < 6 && threadIdx.x != 2 && threadIdx.x != 4) ctl[threadIdx.x] = 0; }

// The whole prepass as ONE workgroup, for launches of at most PRE_FUSED_MAX accesses (small batches: there the ten
// launches above cost more on the host and in launch gaps than the work itself).  Same steps, same arrays, same
// results; the steps are separated by workgroup barriers instead of kernel boundaries.
constexpr int PRE_FUSED_MAX = 4096;      // (round 4: was 12288 -- a C5 batch's 12 288 accesses took 412 us in one workgroup,
                                        //  110 us as ten launches)
constexpr int PRE_THREADS = 1024;
__global__ __launch_bounds__(PRE_THREADS) void k_prepass_fused(
    const int *__restrict__ nodes, const long long *__restrict__ eidx, long long role_stride, int B, int n_roles,
    long long N, int *cnt, int *slot, int *off, int *list, int *wo, int *pflag, int *nxt, int *ctl, int *latch,
    int *hot_node, int *hot_cnt, int *chain_of, int *chain_node, int *chain_len, int *chain_edges, int *owner_of,
    int *hv, int max_chains, int big_min, int grid, int n_models)
{
    __shared__ int sort_s[DEPS_SORT_MAX], sort_t[2 * PRE_THREADS];
    const int tid = threadIdx.x, A = B * n_roles;
    if (tid == 0) { ctl[2] = 0; ctl[13] = 0; }                                   // k_plan_begin
    __syncthreads();
    for (int a = tid; a < A; a += PRE_THREADS) d_count(a, nodes, eidx, role_stride, B, n_roles, N, cnt, slot, ctl, latch);
    __syncthreads();
    for (int a = tid; a < A; a += PRE_THREADS) d_reserve(a, nodes, role_stride, B, n_roles, cnt, off, slot, ctl, hot_node, hot_cnt, big_min);
    __syncthreads();
    for (int a = tid; a < A; a += PRE_THREADS) d_fill(a, nodes, role_stride, B, n_roles, off, slot, list);
    __syncthreads();
    {
        const int n_big = __hip_atomic_load(&ctl[5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int a = tid; a < A; a += PRE_THREADS) d_deps(a, nodes, role_stride, B, n_roles, cnt, off, slot, list, wo, pflag, nxt, n_big, big_min);
        if (n_big <= MAX_BIG)
            for (int g = 0; g < n_big; ++g)
                d_deps_group(tid, PRE_THREADS, hot_node[MAX_HOT + g], B, cnt, off, list, wo, pflag, nxt, sort_s, sort_t);
    }
    d_hot_select(tid, ctl, hot_node, hot_cnt, chain_of, chain_node, chain_len, max_chains, grid, B * n_models, n_models);
    __syncthreads();
    for (int i = tid; i < B; i += PRE_THREADS) d_own(i, nodes, role_stride, B, cnt, slot, wo, chain_of, chain_len, chain_edges, owner_of);
    __syncthreads();
    for (int a = tid; a < A; a += PRE_THREADS) d_hubacc(a, nodes, role_stride, B, n_roles, slot, wo, chain_of, chain_len, hv);
    __syncthreads();
    for (int a = tid; a < (A > MAX_HOT ? A : MAX_HOT); a += PRE_THREADS)
        d_cleanup(a, nodes, role_stride, B, n_roles, slot, cnt, ctl, hot_node, chain_of);
    __syncthreads();
    if (tid < 6 && tid != 2 && tid != 4) ctl[tid] = 0;                           // k_reset_ctl
}


#ifdef ZT_PREPASS_COOP_VARIANT
// the whole prepass as ONE cooperative kernel: measured no faster, kept out of the product (tools/exp/variants/tppr_prepass_coop.hpp)
#include "tppr_prepass_coop.hpp"
#endif

}  // namespace

bool zt::tppr_prepass_coop_compiled()
{
#ifdef ZT_PREPASS_COOP_VARIANT
    return true;
#else
    return false;
#endif
}

// CUs a stream may use (CU-masked streams: the size of the mask)
int zt::tppr_stream_cus(const zt_tppr *h, hipStream_t s)
{
    uint32_t mask[32] = {0};
    if (hipExtStreamGetCUMask(s, 32, mask) != hipSuccess) { (void)hipGetLastError(); return h->n_cu; }
    int c = 0;
    for (int q = 0; q < 32; ++q) c += __builtin_popcount(mask[q]);
    return (c > 0 && c < h->n_cu) ? c : h->n_cu;
}

// the stream a pipeline will launch k_stream on is known before the first plan: plan for ITS compute units (round-4 advisor:
// the first plans of a pipeline assumed the whole chip and planned more chains than the masked stream's grid wants)
void zt::tppr_hint_cus(zt_tppr *h, hipStream_t s) { if (h) h->run_cus = tppr_stream_cus(h, s); }

// hub chains a grid can carry: at most two thirds of its workgroups, so the general queue always keeps waves.  Chain
// workgroups wait for each other's versions and for the general queue's partner tasks: they need the WHOLE grid resident,
// which only a process that has the stream's CUs to itself can promise (share > 1: no chains; ZT_STREAM_CHAINS=0: none either).
int zt::tppr_chains_for_grid(const zt_tppr *h, int grid, int n_models)
{
    static const int chains_env = getenv("ZT_STREAM_CHAINS") ? atoi(getenv("ZT_STREAM_CHAINS")) : MAX_CHAINS;
    if (h->share > 1) return 0;
    int max_chains = (2 * grid) / (3 * n_models);
    if (max_chains > chains_env) max_chains = chains_env;
    if (max_chains > MAX_CHAINS) max_chains = MAX_CHAINS;
    return max_chains < 0 ? 0 : max_chains;
}

// grid of k_stream and the number of hub chains for a launch of B edges on a stream of `cus` CUs
void zt::tppr_launch_shape(const zt_tppr *h, int cus, int B, int n_models, int *grid_out, int *max_chains_out)
{
    long long waves = (long long)B * n_models;
    // The kernel is latency-bound (waves mostly sleep on their predecessors): a few waves per CU
    // drain the independent tasks fast enough, and leave LDS / issue slots to a concurrently
    // running aggregation kernel: one workgroup (8 waves) per CU.
    constexpr double wgs_per_cu = 1.0;
    // Every workgroup of the grid must be resident at once (chain workgroups wait on each other's rows):
    // the CUs of the stream that runs k_stream (a CU-masked stream offers fewer) times the workgroups one
    // CU holds (asked from the runtime at create time).
    // (processes sharing the device -- a rehearsal of several ranks on one GPU -- take their share of the CUs each, so that
    //  the grids of all of them are resident together)
    const int share = h->share > 1 ? h->share : 1;
    if (cus / share >= 1) cus /= share; else cus = 1;
    long long max_waves = (long long)(cus * TPPR_WAVES_PER_WG * wgs_per_cu);
    const long long resident = (long long)cus * (h->wg_per_cu > 0 ? h->wg_per_cu : 1) * TPPR_WAVES_PER_WG;
    if (max_waves > resident) max_waves = resident;
    if (max_waves < TPPR_WAVES_PER_WG) max_waves = TPPR_WAVES_PER_WG;
    if (waves > max_waves) waves = max_waves;
    const int grid = (int)((waves + TPPR_WAVES_PER_WG - 1) / TPPR_WAVES_PER_WG);
    *grid_out = grid;
    // the chain hand-off (two-stage mailbox) is built on the register-resident merge: k <= 30
    *max_chains_out = h->k <= TPPR_REG_K_MAX ? tppr_chains_for_grid(h, grid, n_models) : 0;
}

// The dependency prepass of one launch into plan set q, on stream s.  It reads only the node and
// edge ids, never the T-PPR rows, so it may run while k_stream works on the other set.
int zt::tppr_plan_chunk(zt_tppr *h, int q, const int32_t *nodes, const long long *eidx, long long role_stride, int B,
                        int n_roles, int model, hipStream_t s)
{
    zt_tppr::PlanSet &P = h->set[q];
    if (P.used) ZT_HIP(hipStreamWaitEvent(s, P.consumed, 0));      // k_stream of two calls ago has let go of it
    use_set(h, q);
    const int A = B * n_roles;
    const int tb = 256, gb = (A + tb - 1) / tb;
    const int n_models = model < 0 ? h->M : 1;
    int grid, max_chains;
    tppr_launch_shape(h, h->run_cus > 0 ? h->run_cus : h->n_cu, B, n_models, &grid, &max_chains);
    const int big_min = BIG_MIN;               // group size from which the dependencies come from the cooperative sort
    const int budget_grid = grid;              // chains are budgeted against the general queue's load (d_chain_budget)
    if (A <= PRE_FUSED_MAX) {
        ZT_PROF_BEGIN(s, P_PREPASS);
        k_prepass_fused<<<1, PRE_THREADS, 0, s>>>(nodes, eidx, role_stride, B, n_roles, h->N, h->cnt, h->slot, h->off, h->list,
                                                  h->wo, h->pflag, h->nxt, h->ctl, h->latch_dev, h->hot_node, h->hot_cnt,
                                                  h->chain_of, h->chain_node, h->chain_len, h->chain_edges, h->owner_of,
                                                  h->hv, max_chains, big_min, budget_grid, n_models);
        ZT_PROF_END(s, P_PREPASS);
#ifdef ZT_PREPASS_COOP_VARIANT
    } else if (zt::kernel_choice(ZT_CHOICE_TPPR_PREPASS) == ZT_PREPASS_COOP) {       // (measured no faster, and its fences cost the aggregation: on request)
        ZT_PROF_BEGIN(s, P_PREPASS);
        k_prepass_coop<<<COOP_WGS, COOP_THREADS, 0, s>>>(nodes, eidx, role_stride, B, n_roles, h->N, h->cnt, h->slot, h->off, h->list,
                                                         h->wo, h->pflag, h->nxt, h->ctl, h->latch_dev, h->hot_node, h->hot_cnt,
                                                         h->chain_of, h->chain_node, h->chain_len, h->chain_edges, h->owner_of,
                                                         h->hv, max_chains, big_min, budget_grid, n_models, P.bar_base);
        P.bar_base += (unsigned)COOP_BARRIERS * COOP_WGS;
        ZT_PROF_END(s, P_PREPASS);
#endif
    } else {
        ZT_PROF_BEGIN(s, P_PREPASS);
        k_plan_begin<<<1, 64, 0, s>>>(h->ctl);
        k_count<<<gb, tb, 0, s>>>(nodes, eidx, role_stride, B, n_roles, h->N, h->cnt, h->slot, h->ctl, h->latch_dev);
        k_reserve<<<gb, tb, 0, s>>>(nodes, role_stride, B, n_roles, h->cnt, h->off, h->slot, h->ctl, h->hot_node, h->hot_cnt, big_min);
        k_fill<<<gb, tb, 0, s>>>(nodes, role_stride, B, n_roles, h->off, h->slot, h->list);
        k_deps<<<gb, tb, 0, s>>>(nodes, role_stride, B, n_roles, h->cnt, h->off, h->slot, h->list, h->wo, h->pflag, h->nxt, h->ctl, big_min);
        k_deps_big<<<64, DEPS_BIG_THREADS, 0, s>>>(B, h->cnt, h->off, h->list, h->wo, h->pflag, h->nxt, h->ctl, h->hot_node + MAX_HOT);
        k_hot_select<<<1, MAX_HOT, 0, s>>>(h->ctl, h->hot_node, h->hot_cnt, h->chain_of, h->chain_node, h->chain_len,
                                           max_chains, budget_grid, B * n_models, n_models);
        k_own<<<(B + tb - 1) / tb, tb, 0, s>>>(nodes, role_stride, B, h->cnt, h->slot, h->wo, h->chain_of, h->chain_len,
                                               h->chain_edges, h->owner_of);
        k_hubacc<<<gb, tb, 0, s>>>(nodes, role_stride, B, n_roles, h->slot, h->wo, h->chain_of, h->chain_len, h->hv);
        ZT_PROF_END(s, P_PREPASS);
        // per-node counters and the control words back to their rest state: the set is ready for k_stream
        ZT_PROF_BEGIN(s, P_CLEANUP);
        k_cleanup<<<gb, tb, 0, s>>>(nodes, role_stride, B, n_roles, h->slot, h->cnt, h->ctl, h->hot_node, h->chain_of);
        k_reset_ctl<<<1, 64, 0, s>>>(h->ctl);
        ZT_PROF_END(s, P_CLEANUP);
    }
    ZT_LAUNCH_CHECK();
    ZT_HIP(hipEventRecord(P.planned, s));
    P.nodes = nodes; P.B = B; P.n_roles = n_roles; P.model = model; P.grid = grid; P.max_chains = max_chains;
    return ZT_OK;
}


extern "C" int zt_tppr_plan(zt_tppr *h, const int32_t *nodes_dev, const int64_t *eidx_dev, int64_t B, int32_t n_roles,
                            int32_t model, uint64_t *token_out, void *stream)
{
    if (token_out) *token_out = 0;
    if (!h || B < 0 || (n_roles != 2 && n_roles != 3) || model >= h->M || !token_out) {
        set_error("zt_tppr_plan: bad argument");
        return ZT_ERR_ARG;
    }
    if (int st = latched(h, "zt_tppr_plan")) return st;
    if (B == 0 || B > MAX_CHUNK) return ZT_OK;      // nothing to prepare / a multi-launch call plans inline
    if (h->k > ZT_MAX_K) return ZT_OK;              // (dictionaries wider than a wavefront: tppr_wide.hpp has no prepass)
    if (!nodes_dev || !eidx_dev) { set_error("zt_tppr_plan: NULL buffer"); return ZT_ERR_ARG; }
    const int q = h->next_set;
    h->next_set ^= 1;
    h->set[q].valid = false;
    int rc = zt::tppr_plan_chunk(h, q, nodes_dev, reinterpret_cast<const long long *>(eidx_dev), B, (int)B, n_roles, model,
                        (hipStream_t)stream);
    if (rc != ZT_OK) return rc;
    h->set[q].valid = true;
    h->set[q].token = ++h->plan_serial;
    *token_out = h->set[q].token;
    return ZT_OK;
}

