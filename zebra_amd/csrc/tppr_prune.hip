// Static adjacency (CSR) + pruning T-PPR ("PPI", reference utils/util.py:90-276).
// Compile with -ffp-contract=off (bit-exact float64, see tppr_stream.hip).
//
// One wavefront per query row, all (alpha, beta) models in one walk.  The BFS
// frontier and the insertion-ordered candidate list live in LDS; find_before is
// a P-ary search by P lanes per frontier entry; the CSR tails of a whole level
// are read by one lane per (entry, z) pair, most recent first.  Duplicate
// states are merged keeping the reference's dictionary order (first
// occurrence) and its left-to-right float64 summation order; selection uses
// the exact numba argsort semantics.
#include "numba_sort.hpp"

#include <algorithm>
#include <cstring>
#include <vector>

using namespace zt;

struct zt_csr {
    int64_t N, E2;
    long long *indptr;  // device [N+1]
    int *nbr, *eid;     // device [2E]
    double *ts;         // device [2E]
    // host mirror (find_before on the host side of the shim, tests)
    std::vector<long long> h_indptr;
    std::vector<int> h_nbr, h_eid;
    std::vector<double> h_ts;
};

namespace {

constexpr int PR_WAVES = 4;           // queries per workgroup
constexpr int MAX_CAND = 1280;        // sum_{d<=depth} width^d
constexpr int MAX_FRONT = 512;        // width^(depth-1)
constexpr int PR_MAX_MODELS = 4;      // (alpha, beta) models sharing one walk; more run as several launches

struct PruneModels {
    int M;
    double alpha[PR_MAX_MODELS], beta[PR_MAX_MODELS];
};

// Per-wave LDS block, carved from dynamic shared memory; sized at launch from
// the actual (width, depth, models) so that small configurations keep occupancy high.
struct PruneLds {
    u64 *key;               // [cap_c] candidate states in BFS (= dictionary insertion) order
    double *ts;             // [cap_c]
    double *w;              // [M][cap_c] weight of every occurrence, per model
    int *perm;              // [cap_c] owner frontier entry of a new state (walk) / first occurrence (merge) / sort scratch
    int *sel;               // 64 (k <= ZT_MAX_K) or 256
    int *stk;               // 96
    int *f_cnt;             // [cap_f] per frontier entry: number of states it emits
    int *f_off;             // [cap_f] exclusive scan of f_cnt
    int *f_ngh;             // [cap_f] find_before count
    long long *f_lo;        // [cap_f] start of the entry's adjacency
    double *f_base;         // [M][cap_f] weight of the entry's most recent neighbour
    SortLds *sort;          // wave-parallel exact argsort scratch (n <= 128)
};

__host__ __device__ inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

// (sel: 64 entries for k <= ZT_MAX_K -- the tuned configurations keep their LDS footprint --, 256 for the wider k)
__host__ __device__ inline size_t prune_sel_words(int k) { return k <= 64 ? 64 : 256; }
__host__ __device__ inline size_t prune_lds_bytes(int cap_c, int cap_f, int M, int k)
{
    return (2 + (size_t)M) * align16((size_t)cap_c * 8) + align16((size_t)cap_c * 4) + align16(prune_sel_words(k) * 4) + align16(96 * 4) +
           3 * align16((size_t)cap_f * 4) + (1 + (size_t)M) * align16((size_t)cap_f * 8) + align16(sizeof(SortLds));
}

__device__ inline PruneLds carve(char *base, int cap_c, int cap_f, int M, int k)
{
    PruneLds L;
    char *p = base;
    L.key = reinterpret_cast<u64 *>(p); p += align16((size_t)cap_c * 8);
    L.ts = reinterpret_cast<double *>(p); p += align16((size_t)cap_c * 8);
    L.w = reinterpret_cast<double *>(p); p += (size_t)M * align16((size_t)cap_c * 8);
    L.perm = reinterpret_cast<int *>(p); p += align16((size_t)cap_c * 4);
    L.sel = reinterpret_cast<int *>(p); p += align16(prune_sel_words(k) * 4);
    L.stk = reinterpret_cast<int *>(p); p += align16(96 * 4);
    L.f_cnt = reinterpret_cast<int *>(p); p += align16((size_t)cap_f * 4);
    L.f_off = reinterpret_cast<int *>(p); p += align16((size_t)cap_f * 4);
    L.f_ngh = reinterpret_cast<int *>(p); p += align16((size_t)cap_f * 4);
    L.f_lo = reinterpret_cast<long long *>(p); p += align16((size_t)cap_f * 8);
    L.f_base = reinterpret_cast<double *>(p); p += (size_t)M * align16((size_t)cap_f * 8);
    L.sort = reinterpret_cast<SortLds *>(p);
    return L;
}

// numba pow(float64, int64) (numba/cpython/numbers.py:207-243)
__device__ __forceinline__ double numba_int_pow(double a, long long b)
{
    if (b > 0x10000) return pow(a, (double)b);
    double r = 1.0;
    long long e = b;
    while (e != 0) {
        if (e & 1) r *= a;
        e >>= 1;
        a *= a;
    }
    return r;
}

// NeighborFinder.get_pruned_topk (utils/util.py:185-276) for every (alpha, beta) model at once: ONE wavefront per
// query row walks the adjacency once -- which states are reached depends on (node, time) only -- and carries one
// weight per model along.
//   find_before (np.searchsorted, :152-154): a P-ary search by P lanes per frontier entry (P = 64 for the single
//     entry of level 0, 8 when a level has many) -- log_P(degree) dependent round trips instead of log_2, once per
//     entry;
//   the <= width most recent neighbours of all entries of a level (:211-232): one lane per (entry, z) pair, so the
//     tails come in as one coalesced round of loads;
//   duplicate states (dict[state] += w in occurrence order, :222-225), exact numba argsort selection (:240-276)
//     per model.
__global__ __launch_bounds__(WAVE * PR_WAVES) void k_pruned_topk(
    const long long *__restrict__ indptr, const int *__restrict__ nbr, const int *__restrict__ eid,
    const double *__restrict__ ats, long long num_nodes, const int *__restrict__ q_nodes,
    const double *__restrict__ q_ts, long long nq, int width, int depth, PruneModels pm, int k,
    int *out_nodes, int *out_eidx, float *out_dt, float *out_w, long long out_stride, int *status, int cap_c, int cap_f,
    int dbg_stop, int zero_empty)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int M = pm.M;
    const PruneLds L = carve(smem + (threadIdx.x / WAVE) * prune_lds_bytes(cap_c, cap_f, M, k), cap_c, cap_f, M, k);
    const size_t wst = align16((size_t)cap_c * 8) / 8, bst = align16((size_t)cap_f * 8) / 8;   // model strides (doubles)
    const int lane = lane_id();
    const long long qi = (long long)blockIdx.x * PR_WAVES + threadIdx.x / WAVE;
    if (qi >= nq) return;
    const int qn = q_nodes[qi];
    const double qt = q_ts[qi];
    if (qn < 0 || qn >= num_nodes) {
        if (lane == 0) atomicExch(status, ZT_ERR_RANGE);
        // (zero_empty: nobody cleared the slot -- the row would keep the previous group's neighbours and the aggregation of
        //  this step, whose caller may never look at the status word, would consume them: an empty row instead)
        if (zero_empty)
            for (int m = 0; m < M; ++m) {
                const long long ob = (long long)m * out_stride + qi * k;
                for (int j = lane; j < k; j += WAVE) { out_nodes[ob + j] = 0; out_eidx[ob + j] = 0; out_w[ob + j] = 0.f; out_dt[ob + j] = 0.f; }
            }
        return;
    }

    // frontier of level `dep` = the states level dep-1 appended: candidates [fr_lo, fr_lo + nf); level 0: the query
    int n_cand = 0, fr_lo = 0, nf = 1;
    for (int dep = 0; dep < depth; ++dep) {
        // ---- find_before of every frontier entry: P lanes per entry ----
        const int P = nf == 1 ? 64 : (nf == 2 ? 32 : (nf <= 4 ? 16 : 8));
        const int epc = WAVE / P;                                   // entries per pass
        const int g = lane / P, j = lane % P;
        const u64 gmask = P == 64 ? ~0ull : ((1ull << P) - 1ull);
        for (int f0 = 0; f0 < nf; f0 += epc) {
            const int f = f0 + g;
            const bool act = f < nf;
            const int node = !act ? 0 : (dep == 0 ? qn : (int)(unsigned)(L.key[fr_lo + f] & 0xffffffffull));
            const double t = !act ? 0.0 : (dep == 0 ? qt : L.ts[fr_lo + f]);
            const long long lo0 = act ? indptr[node] : 0, hi0 = act ? indptr[node + 1] : 0;
            long long lo = lo0, hi = hi0;                           // the answer (first index with ts >= t) is in [lo, hi]
            while (__ballot(lo < hi) != 0ull) {
                const long long n = hi - lo;
                const bool open = lo < hi;
                const bool pred = open && ats[lo + (n * j) / P] < t;               // probes at lo + floor(n * j / P)
                const int c = __popcll((__ballot(pred) >> (g * P)) & gmask);       // true for a prefix of the probes
                if (open) {
                    if (c == 0) hi = lo;
                    else {
                        const long long nlo = lo + (n * (c - 1)) / P + 1;
                        hi = c < P ? lo + (n * c) / P : hi;
                        lo = nlo;
                    }
                }
            }
            if (act && j == 0) {
                const long long n_ngh = lo - lo0;
                L.f_ngh[f] = (int)n_ngh;                            // < 2^31: entries of one node
                L.f_lo[f] = lo0;
                L.f_cnt[f] = (int)(n_ngh < width ? n_ngh : width);
            }
        }
        wave_sync();
        // ---- exclusive scan of f_cnt ----
        int n_new = 0;
        for (int f0 = 0; f0 < nf; f0 += WAVE) {
            const int f = f0 + lane;
            const int c = f < nf ? L.f_cnt[f] : 0;
            int inc = c;
#pragma unroll
            for (int d = 1; d < WAVE; d <<= 1) {
                const int o = __shfl_up(inc, d);
                if (lane >= d) inc += o;
            }
            if (f < nf) L.f_off[f] = n_new + inc - c;
            n_new += __shfl(inc, WAVE - 1);
        }
        if (n_new == 0) break;                                      // :234-235
        // ---- per entry and model: weight of its most recent neighbour (:208-209); who owns which new state ----
        for (int f0 = 0; f0 < nf; f0 += WAVE) {
            const int f = f0 + lane;
            if (f < nf) {
                const int c = L.f_cnt[f], o = n_cand + L.f_off[f];
                if (c > 0) {
                    const long long n_ngh = L.f_ngh[f];
                    for (int m = 0; m < M; ++m) {
                        const double alpha = pm.alpha[m], beta = pm.beta[m];
                        const double qw = dep == 0 ? 1.0 : L.w[m * wst + fr_lo + f];
                        const double norm = beta / (1.0 - beta) * (1.0 - numba_int_pow(beta, n_ngh));   // :208
                        L.f_base[m * bst + f] = (alpha != 0.0 && dep == 0) ? qw * (1.0 - alpha) * beta / norm * alpha
                                                                           : qw * (1.0 - alpha) * beta / norm;   // :209
                    }
                    for (int z = 0; z < c; ++z) L.perm[o + z] = f;
                }
            }
        }
        wave_sync();
        // ---- the new states, most recent first (:211-232): one lane per (entry, z) ----
        for (int i0 = 0; i0 < n_new; i0 += WAVE) {
            const int i = i0 + lane;
            if (i < n_new) {
                const int f = L.perm[n_cand + i];
                const int z = i - L.f_off[f];
                const long long p = L.f_lo[f] + L.f_ngh[f] - (z + 1);
                L.key[n_cand + i] = ((u64)(unsigned)eid[p] << 32) | (u64)(unsigned)nbr[p];
                L.ts[n_cand + i] = ats[p];
                for (int m = 0; m < M; ++m) {
                    const double beta = pm.beta[m];
                    double weight = L.f_base[m * bst + f];
                    for (int q = 0; q < z; ++q) weight = weight * beta;             // weight *= beta after every state
                    L.w[m * wst + n_cand + i] = weight;
                }
            }
        }
        wave_sync();
        fr_lo = n_cand;
        nf = n_new;
        n_cand += n_new;
    }
    if (n_cand == 0) {                                              // :241-242, row untouched
        // (zero_empty: the caller's output arrays are not cleared beforehand -- pipeline.hip: a memset in front of every
        //  query is a packet on the T-PPR stream, ~6 us of every C4 step -- so an empty row is written here, as zeros)
        if (zero_empty)
            for (int m = 0; m < M; ++m) {
                const long long ob = (long long)m * out_stride + qi * k;
                for (int j = lane; j < k; j += WAVE) { out_nodes[ob + j] = 0; out_eidx[ob + j] = 0; out_w[ob + j] = 0.f; out_dt[ob + j] = 0.f; }
            }
        return;
    }
    if (dbg_stop == 1) { if (lane == 0) out_nodes[qi * k] = n_cand; return; }      // (diagnostic: ZT_PRUNE_STOP)

    // ---- merge duplicate states: dict[state] += weight in occurrence order (:222-225) ----
    // perm[c] = first occurrence of c's state (= c for a leader); every lane walks the list front to back with
    // broadcast reads.  The frontier above needed the un-merged weights; from here on only leaders matter.
    bool any_dup = false;
    for (int c0 = 0; c0 < n_cand; c0 += WAVE) {
        const int c = c0 + lane;
        const bool in = c < n_cand;
        const u64 kc = in ? L.key[c] : 0ull;
        const double tc = in ? L.ts[c] : 0.0;
        int fi = c;
        const int qmax = c0 + WAVE - 1 < n_cand ? c0 + WAVE - 1 : n_cand - 1;      // nobody looks beyond its own index
        // first earlier entry with the same (edge, node): four broadcast reads in flight per step
        for (int q0 = 0; q0 < qmax; q0 += 4) {
            u64 kq[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) kq[u] = L.key[q0 + u < n_cand ? q0 + u : n_cand - 1];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (in && q0 + u < c && fi == c && kq[u] == kc) fi = q0 + u;
        }
        // the dictionary key is (edge, node, time): equal (edge, node) with different times cannot come out of one
        // adjacency, but if it ever does, redo this chunk comparing the time on every hit
        if (__ballot(in && fi != c && L.ts[fi] != tc) != 0ull) {
            fi = c;
            for (int q = 0; q < qmax; ++q)
                if (in && q < c && fi == c && L.key[q] == kc && L.ts[q] == tc) fi = q;
        }
        if (in) L.perm[c] = fi;
        any_dup = any_dup || __ballot(in && fi != c) != 0ull;
    }
    wave_sync();
    int nd = n_cand;
    if (any_dup) {
        // a leader's value is the left-to-right sum of its occurrences: later occurrences are added in index order,
        // lane m doing model m (a lane's LDS accesses execute in program order)
        for (int c0 = 0; c0 < n_cand; c0 += WAVE) {
            const int c = c0 + lane;
            u64 dm = __ballot(c < n_cand && L.perm[c] != c);
            while (dm != 0ull) {
                const int cc = c0 + __ffsll((long long)dm) - 1;
                dm &= dm - 1ull;
                const int lead = L.perm[cc];
                if (lane < M) L.w[lane * wst + lead] = L.w[lane * wst + lead] + L.w[lane * wst + cc];
            }
        }
        wave_sync();
        // compact the leaders in order (dictionary insertion order)
        nd = 0;
        for (int c0 = 0; c0 < n_cand; c0 += WAVE) {
            const int c = c0 + lane;
            const bool lead = c < n_cand && L.perm[c] == c;
            const u64 kc = lead ? L.key[c] : 0;
            const double tc = lead ? L.ts[c] : 0.0;
            double wc[PR_MAX_MODELS];
#pragma unroll
            for (int m = 0; m < PR_MAX_MODELS; ++m) wc[m] = (lead && m < M) ? L.w[m * wst + c] : 0.0;
            const u64 bm = __ballot(lead);
            wave_sync();   // all reads of this chunk done before it may be overwritten
            if (lead) {
                const int pos = nd + __popcll(bm & lanemask_lt());   // pos <= c: never clobbers unread chunks
                L.key[pos] = kc; L.ts[pos] = tc;
#pragma unroll
                for (int m = 0; m < PR_MAX_MODELS; ++m) if (m < M) L.w[m * wst + pos] = wc[m];
            }
            nd += __popcll(bm);
            wave_sync();
        }
    }

    if (dbg_stop == 2) { if (lane == 0) out_nodes[qi * k] = nd; return; }
    // ---- select and emit (:240-276), model by model ----
    for (int m = 0; m < M; ++m) {
        const long long ob = (long long)m * out_stride + qi * k;
        const double *wm = L.w + m * wst;
        // (k <= ZT_MAX_K: one pass, lane j = entry j; wider k -- the reference puts no bound on --topk, train.py:46 -- strides)
        if (nd <= k) {
            for (int j = lane; j < k; j += WAVE) {
                const bool a = j < nd;
                out_nodes[ob + j] = a ? (int)(unsigned)(L.key[j] & 0xffffffffull) : 0;
                out_eidx[ob + j] = a ? (int)(unsigned)(L.key[j] >> 32) : 0;
                out_w[ob + j] = a ? (float)wm[j] : 0.f;
                const float tsf = a ? (float)L.ts[j] : 0.f;
                out_dt[ob + j] = (float)(qt - (double)tsf);
            }
            continue;
        }
        if (dbg_stop == 4 || (dbg_stop == 3 && nd > WAVE)) { if (lane < k) L.sel[lane] = lane; wave_sync(); }   // (diagnostic)
        else if (k <= ZT_MAX_K) topk_select_wave(wm, nd, k, L.sel, *L.sort, L.perm, L.stk);
        else topk_select_any(wm, nd, k, L.sel, *L.sort, L.perm, L.stk);      // kept sets wider than a wavefront: correct first
        for (int j = lane; j < k; j += WAVE) {
            const int c = L.sel[j];
            out_nodes[ob + j] = (int)(unsigned)(L.key[c] & 0xffffffffull);
            out_eidx[ob + j] = (int)(unsigned)(L.key[c] >> 32);
            out_w[ob + j] = (float)wm[c];
            out_dt[ob + j] = (float)(qt - (double)(float)L.ts[c]);
        }
        wave_sync();
    }
}

}  // namespace

static int csr_upload(zt_csr *c)
{
    const int64_t n2 = c->E2;
    ZT_HIP(hipMalloc(&c->indptr, sizeof(long long) * (c->N + 1)));
    ZT_HIP(hipMalloc(&c->nbr, sizeof(int) * (n2 > 0 ? n2 : 1)));
    ZT_HIP(hipMalloc(&c->eid, sizeof(int) * (n2 > 0 ? n2 : 1)));
    ZT_HIP(hipMalloc(&c->ts, sizeof(double) * (n2 > 0 ? n2 : 1)));
    ZT_HIP(hipMemcpy(c->indptr, c->h_indptr.data(), sizeof(long long) * (c->N + 1), hipMemcpyHostToDevice));
    if (n2 > 0) {
        ZT_HIP(hipMemcpy(c->nbr, c->h_nbr.data(), sizeof(int) * n2, hipMemcpyHostToDevice));
        ZT_HIP(hipMemcpy(c->eid, c->h_eid.data(), sizeof(int) * n2, hipMemcpyHostToDevice));
        ZT_HIP(hipMemcpy(c->ts, c->h_ts.data(), sizeof(double) * n2, hipMemcpyHostToDevice));
    }
    return ZT_OK;
}

extern "C" int zt_csr_build(zt_csr **out, const int32_t *src, const int32_t *dst, const int64_t *eidx,
                            const double *ts, int64_t E, int64_t num_nodes)
{
    if (!out || E < 0 || num_nodes <= 0 || (E > 0 && (!src || !dst || !eidx || !ts))) {
        set_error("zt_csr_build: bad argument");
        return ZT_ERR_ARG;
    }
    for (int64_t i = 0; i < E; ++i) {
        if (src[i] < 0 || src[i] >= num_nodes || dst[i] < 0 || dst[i] >= num_nodes || eidx[i] < 0 ||
            eidx[i] > 0x7fffffffll) {
            set_error("zt_csr_build: id out of range at edge %lld", (long long)i);
            return ZT_ERR_RANGE;
        }
    }
    zt_csr *c = new zt_csr();
    c->N = num_nodes;
    c->E2 = 2 * E;
    c->h_indptr.assign(num_nodes + 1, 0);
    for (int64_t i = 0; i < E; ++i) { c->h_indptr[src[i] + 1]++; c->h_indptr[dst[i] + 1]++; }
    for (int64_t v = 0; v < num_nodes; ++v) c->h_indptr[v + 1] += c->h_indptr[v];
    // adjacency in stream order (both directions per edge, utils/util.py:94-96),
    // then a stable sort by timestamp per node (:103)
    std::vector<long long> cur(c->h_indptr.begin(), c->h_indptr.end() - 1);
    std::vector<int64_t> pos(2 * E);            // slot -> (edge index * 2 + direction)
    for (int64_t i = 0; i < E; ++i) { pos[cur[src[i]]++] = 2 * i; pos[cur[dst[i]]++] = 2 * i + 1; }
    for (int64_t v = 0; v < num_nodes; ++v)
        std::stable_sort(pos.begin() + c->h_indptr[v], pos.begin() + c->h_indptr[v + 1],
                         [&](int64_t a, int64_t b) { return ts[a >> 1] < ts[b >> 1]; });
    c->h_nbr.resize(2 * E); c->h_eid.resize(2 * E); c->h_ts.resize(2 * E);
    for (int64_t p = 0; p < 2 * E; ++p) {
        const int64_t i = pos[p] >> 1;
        c->h_nbr[p] = (pos[p] & 1) ? src[i] : dst[i];
        c->h_eid[p] = (int)eidx[i];
        c->h_ts[p] = ts[i];
    }
    {
        int rc = csr_upload(c);
        if (rc != ZT_OK) { delete c; return rc; }
    }
    *out = c;
    return ZT_OK;
}

extern "C" int zt_csr_from_sorted(zt_csr **out, const int64_t *indptr, const int32_t *nbr, const int32_t *eid,
                                  const double *ts, int64_t num_nodes)
{
    if (!out || !indptr || num_nodes <= 0) { set_error("zt_csr_from_sorted: bad argument"); return ZT_ERR_ARG; }
    const int64_t n2 = indptr[num_nodes];
    if (indptr[0] != 0 || n2 < 0) { set_error("zt_csr_from_sorted: bad indptr"); return ZT_ERR_ARG; }
    for (int64_t v = 0; v < num_nodes; ++v)
        if (indptr[v + 1] < indptr[v]) { set_error("zt_csr_from_sorted: indptr not monotone"); return ZT_ERR_ARG; }
    for (int64_t p = 0; p < n2; ++p)
        if (nbr[p] < 0 || nbr[p] >= num_nodes || eid[p] < 0) {
            set_error("zt_csr_from_sorted: id out of range at entry %lld", (long long)p);
            return ZT_ERR_RANGE;
        }
    zt_csr *c = new zt_csr();
    c->N = num_nodes;
    c->E2 = n2;
    c->h_indptr.assign(indptr, indptr + num_nodes + 1);
    c->h_nbr.assign(nbr, nbr + n2);
    c->h_eid.assign(eid, eid + n2);
    c->h_ts.assign(ts, ts + n2);
    int rc = csr_upload(c);
    if (rc != ZT_OK) { delete c; return rc; }
    *out = c;
    return ZT_OK;
}

extern "C" int zt_csr_size(const zt_csr *c, int64_t *num_nodes, int64_t *num_entries)
{
    if (!c) return ZT_ERR_ARG;
    if (num_nodes) *num_nodes = c->N;
    if (num_entries) *num_entries = c->E2;
    return ZT_OK;
}

extern "C" int zt_csr_export(const zt_csr *c, int64_t *indptr, int32_t *nbr, int32_t *eid, double *ts)
{
    if (!c) return ZT_ERR_ARG;
    if (indptr) for (int64_t v = 0; v <= c->N; ++v) indptr[v] = c->h_indptr[v];
    if (nbr) memcpy(nbr, c->h_nbr.data(), sizeof(int) * c->E2);
    if (eid) memcpy(eid, c->h_eid.data(), sizeof(int) * c->E2);
    if (ts) memcpy(ts, c->h_ts.data(), sizeof(double) * c->E2);
    return ZT_OK;
}

extern "C" int zt_csr_destroy(zt_csr *c)
{
    if (!c) return ZT_OK;
    (void)hipFree(c->indptr); (void)hipFree(c->nbr); (void)hipFree(c->eid); (void)hipFree(c->ts);
    delete c;
    return ZT_OK;
}

extern "C" int zt_csr_find_before(const zt_csr *c, int32_t v, double t, int64_t *count, int32_t *nbr_host,
                                  int32_t *eid_host, double *ts_host, int64_t cap)
{
    if (!c || !count) return ZT_ERR_ARG;
    if (v < 0 || v >= c->N) { set_error("zt_csr_find_before: node id out of range"); return ZT_ERR_RANGE; }
    const long long lo = c->h_indptr[v], hi = c->h_indptr[v + 1];
    const long long n = std::lower_bound(c->h_ts.begin() + lo, c->h_ts.begin() + hi, t) - (c->h_ts.begin() + lo);
    *count = n;
    const long long m = n < cap ? n : cap;
    if (nbr_host) memcpy(nbr_host, c->h_nbr.data() + lo, sizeof(int) * m);
    if (eid_host) memcpy(eid_host, c->h_eid.data() + lo, sizeof(int) * m);
    if (ts_host) memcpy(ts_host, c->h_ts.data() + lo, sizeof(double) * m);
    return ZT_OK;
}

// one launch for up to PR_MAX_MODELS models; out arrays are [M][nq][k]
static int pruned_launch(const zt_csr *c, const int32_t *q_nodes_dev, const double *q_ts_dev, int64_t nq, int32_t width,
                         int32_t depth, int M, const double *alpha, const double *beta, int32_t k, int32_t *on, int32_t *oe,
                         float *od, float *ow, int32_t *status_dev, hipStream_t s, bool zero_empty = false)
{
    long long cap = 0, lvl = 1, front = 1;
    for (int d = 0; d < depth; ++d) { front = lvl; lvl *= width; cap += lvl; if (cap > MAX_CAND) break; }
    if (k > ZT_MAX_K_WIDE || cap > MAX_CAND || front > MAX_FRONT) {
        set_error("zt_pruned_topk: k=%d width=%d depth=%d exceeds the LDS-resident limits "
                  "(k<=%d, sum width^d<=%d)", k, width, depth, ZT_MAX_K_WIDE, MAX_CAND);
        return ZT_ERR_UNSUPPORTED;
    }
    const int cap_c = (int)cap, cap_f = (int)front;
#ifdef ZT_DIAG
    static const int dbg_stop = getenv("ZT_PRUNE_STOP") ? atoi(getenv("ZT_PRUNE_STOP")) : 0;   // diagnostic builds only (WRONG results): 1 walk only, 2 + merge
#else
    constexpr int dbg_stop = 0;
#endif
    for (int m0 = 0; m0 < M;) {
        // as many models per launch as the workgroup's LDS allows (at least one)
        int mm = M - m0 < PR_MAX_MODELS ? M - m0 : PR_MAX_MODELS;
        while (mm > 1 && prune_lds_bytes(cap_c, cap_f, mm, k) * PR_WAVES > 64 * 1024) --mm;
        PruneModels pm;
        pm.M = mm;
        for (int q = 0; q < mm; ++q) { pm.alpha[q] = alpha[m0 + q]; pm.beta[q] = beta[m0 + q]; }
        const size_t lds = prune_lds_bytes(cap_c, cap_f, mm, k) * PR_WAVES;
        static size_t attr_lds = 0;
        if (lds > 48 * 1024 && lds > attr_lds) {
            ZT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_pruned_topk),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_lds = lds;
        }
        const int grid = (int)((nq + PR_WAVES - 1) / PR_WAVES);
        const size_t o = (size_t)m0 * nq * k;
        ZT_PROF_BEGIN(s, P_PRUNE);
        k_pruned_topk<<<grid, WAVE * PR_WAVES, lds, s>>>(c->indptr, c->nbr, c->eid, c->ts, c->N, q_nodes_dev, q_ts_dev, nq,
                                                         width, depth, pm, k, on + o, oe + o, od + o, ow + o,
                                                         (long long)nq * k, status_dev, cap_c, cap_f, dbg_stop, zero_empty ? 1 : 0);
        ZT_PROF_END(s, P_PRUNE);
        ZT_LAUNCH_CHECK();
        m0 += mm;
    }
    return ZT_OK;
}

extern "C" int zt_pruned_topk(const zt_csr *c, const int32_t *q_nodes_dev, const double *q_ts_dev, int64_t nq,
                              int32_t width, int32_t depth, double alpha, double beta, int32_t k,
                              int32_t *out_nodes_dev, int32_t *out_eidx_dev, float *out_dt_dev, float *out_w_dev,
                              int32_t *status_dev, void *stream)
{
    if (!c || nq < 0 || width <= 0 || depth <= 0 || k <= 0 || !status_dev) {
        set_error("zt_pruned_topk: bad argument");
        return ZT_ERR_ARG;
    }
    if (nq == 0) return ZT_OK;
    return pruned_launch(c, q_nodes_dev, q_ts_dev, nq, width, depth, 1, &alpha, &beta, k, out_nodes_dev, out_eidx_dev,
                         out_dt_dev, out_w_dev, status_dev, (hipStream_t)stream);
}

// zt_pruned_topk_multi into output arrays that were NOT cleared: rows with an empty dictionary are written as zeros by the kernel
int zt::pruned_topk_multi_fill(const zt_csr *c, const int32_t *q_nodes_dev, const double *q_ts_dev, int64_t nq, int32_t width,
                               int32_t depth, int32_t n_models, const double *alpha_host, const double *beta_host, int32_t k,
                               int32_t *out_nodes_dev, int32_t *out_eidx_dev, float *out_dt_dev, float *out_w_dev,
                               int32_t *status_dev, void *stream)
{
    if (!c || nq < 0 || width <= 0 || depth <= 0 || k <= 0 || !status_dev || n_models <= 0 || !alpha_host || !beta_host) {
        set_error("zt_pruned_topk_multi: bad argument");
        return ZT_ERR_ARG;
    }
    if (nq == 0) return ZT_OK;
    return pruned_launch(c, q_nodes_dev, q_ts_dev, nq, width, depth, n_models, alpha_host, beta_host, k, out_nodes_dev,
                         out_eidx_dev, out_dt_dev, out_w_dev, status_dev, (hipStream_t)stream, true);
}

extern "C" int zt_pruned_topk_multi(const zt_csr *c, const int32_t *q_nodes_dev, const double *q_ts_dev, int64_t nq,
                                    int32_t width, int32_t depth, int32_t n_models, const double *alpha_host,
                                    const double *beta_host, int32_t k, int32_t *out_nodes_dev, int32_t *out_eidx_dev,
                                    float *out_dt_dev, float *out_w_dev, int32_t *status_dev, void *stream)
{
    if (!c || nq < 0 || width <= 0 || depth <= 0 || k <= 0 || !status_dev || n_models <= 0 || !alpha_host || !beta_host) {
        set_error("zt_pruned_topk_multi: bad argument");
        return ZT_ERR_ARG;
    }
    if (nq == 0) return ZT_OK;
    return pruned_launch(c, q_nodes_dev, q_ts_dev, nq, width, depth, n_models, alpha_host, beta_host, k, out_nodes_dev,
                         out_eidx_dev, out_dt_dev, out_w_dev, status_dev, (hipStream_t)stream);
}
