// Static adjacency (CSR) + pruning T-PPR ("PPI", reference utils/util.py:90-276).
// Compile with -ffp-contract=off (bit-exact float64, see tppr_stream.hip).
//
// One wavefront per query row.  The BFS frontier and the insertion-ordered
// candidate list live in LDS; CSR tails are read most-recent-first straight
// from HBM (each frontier entry reads one contiguous tail of <= width entries).
// Duplicate states are merged with an all-pairs key comparison that keeps the
// reference's dictionary order (first occurrence) and its left-to-right
// float64 summation order; selection uses the exact numba argsort semantics.
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

// Per-wave LDS block, carved from dynamic shared memory; sized at launch from
// the actual (width, depth) so that small configurations keep occupancy high.
struct PruneLds {
    u64 *key;
    double *ts;
    double *w;
    int *perm;      // also: first-occurrence flags
    int *sel;       // 64
    int *stk;       // 96
    int *f_cnt;     // per frontier entry: number of states it emits
    int *f_off;
    SortLds *sort;  // wave-parallel exact argsort scratch (n <= 128)
};

__host__ __device__ inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

__host__ __device__ inline size_t prune_lds_bytes(int cap_c, int cap_f)
{
    return 3 * align16((size_t)cap_c * 8) + align16((size_t)cap_c * 4) + align16(64 * 4) + align16(96 * 4) +
           2 * align16((size_t)cap_f * 4) + align16(sizeof(SortLds));
}

__device__ inline PruneLds carve(char *base, int cap_c, int cap_f)
{
    PruneLds L;
    char *p = base;
    L.key = reinterpret_cast<u64 *>(p); p += align16((size_t)cap_c * 8);
    L.ts = reinterpret_cast<double *>(p); p += align16((size_t)cap_c * 8);
    L.w = reinterpret_cast<double *>(p); p += align16((size_t)cap_c * 8);
    L.perm = reinterpret_cast<int *>(p); p += align16((size_t)cap_c * 4);
    L.sel = reinterpret_cast<int *>(p); p += align16(64 * 4);
    L.stk = reinterpret_cast<int *>(p); p += align16(96 * 4);
    L.f_cnt = reinterpret_cast<int *>(p); p += align16((size_t)cap_f * 4);
    L.f_off = reinterpret_cast<int *>(p); p += align16((size_t)cap_f * 4);
    L.sort = reinterpret_cast<SortLds *>(p);
    return L;
}

__device__ __forceinline__ long long find_before(const double *ats, long long lo, long long hi, double t)
{
    // np.searchsorted(side='left'): first index with ts >= t
    const long long base = lo;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (ats[mid] < t) lo = mid + 1; else hi = mid;
    }
    return lo - base;
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

__global__ __launch_bounds__(WAVE * PR_WAVES) void k_pruned_topk(
    const long long *__restrict__ indptr, const int *__restrict__ nbr, const int *__restrict__ eid,
    const double *__restrict__ ats, long long num_nodes, const int *__restrict__ q_nodes,
    const double *__restrict__ q_ts, long long nq, int width, int depth, double alpha, double beta, int k,
    int *out_nodes, int *out_eidx, float *out_dt, float *out_w, int *status, int cap_c, int cap_f)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const PruneLds L = carve(smem + (threadIdx.x / WAVE) * prune_lds_bytes(cap_c, cap_f), cap_c, cap_f);
    const int lane = lane_id();
    const long long qi = (long long)blockIdx.x * PR_WAVES + threadIdx.x / WAVE;
    if (qi >= nq) return;
    const int qn = q_nodes[qi];
    const double qt = q_ts[qi];
    if (qn < 0 || qn >= num_nodes) {
        if (lane == 0) atomicExch(status, ZT_ERR_RANGE);
        return;
    }

    // frontier of level `dep` = candidates [fr_lo, fr_hi) of the previous level
    // (node = key & 0xffffffff, ts, w); level 0 frontier is the query itself.
    int n_cand = 0;
    int fr_lo = 0, fr_hi = 0;
    for (int dep = 0; dep < depth; ++dep) {
        const int nf = dep == 0 ? 1 : fr_hi - fr_lo;
        // pass 1: how many states does each frontier entry emit?
        for (int f0 = 0; f0 < nf; f0 += WAVE) {
            const int f = f0 + lane;
            int c = 0;
            if (f < nf) {
                const int node = dep == 0 ? qn : (int)(unsigned)(L.key[fr_lo + f] & 0xffffffffull);
                const double t = dep == 0 ? qt : L.ts[fr_lo + f];
                const long long n_ngh = find_before(ats, indptr[node], indptr[node + 1], t);
                c = (int)(n_ngh < width ? n_ngh : width);
                L.f_cnt[f] = c;
            }
        }
        wave_sync();
        // exclusive scan of f_cnt (nf <= MAX_FRONT; one lane, short)
        if (lane == 0) {
            int acc = 0;
            for (int f = 0; f < nf; ++f) { L.f_off[f] = acc; acc += L.f_cnt[f]; }
            L.sel[0] = acc;
        }
        wave_sync();
        const int n_new = L.sel[0];
        wave_sync();
        if (n_new == 0) break;                                   // :234-235
        // pass 2: each frontier entry appends its states, most recent first
        for (int f0 = 0; f0 < nf; f0 += WAVE) {
            const int f = f0 + lane;
            if (f < nf && L.f_cnt[f] > 0) {
                const int node = dep == 0 ? qn : (int)(unsigned)(L.key[fr_lo + f] & 0xffffffffull);
                const double t = dep == 0 ? qt : L.ts[fr_lo + f];
                const double qw = dep == 0 ? 1.0 : L.w[fr_lo + f];
                const long long lo = indptr[node];
                const long long n_ngh = find_before(ats, lo, indptr[node + 1], t);
                const double norm = beta / (1.0 - beta) * (1.0 - numba_int_pow(beta, n_ngh));   // :208
                double weight = (alpha != 0.0 && dep == 0) ? qw * (1.0 - alpha) * beta / norm * alpha
                                                          : qw * (1.0 - alpha) * beta / norm;      // :209
                const int c = L.f_cnt[f];
                const int o = n_cand + L.f_off[f];
                for (int z = 0; z < c; ++z) {                    // :211-232
                    const long long p = lo + n_ngh - (z + 1);
                    L.key[o + z] = ((u64)(unsigned)eid[p] << 32) | (u64)(unsigned)nbr[p];
                    L.ts[o + z] = ats[p];
                    L.w[o + z] = weight;
                    weight = weight * beta;
                }
            }
        }
        wave_sync();
        fr_lo = n_cand;
        n_cand += n_new;
        fr_hi = n_cand;
    }
    if (n_cand == 0) return;                                     // :241-242, row untouched

    // ---- merge duplicate states: dict[state] += weight in occurrence order ----
    // perm[c] = 1 if c is the first occurrence of its key.
    for (int c = lane; c < n_cand; c += WAVE) {
        const u64 kc = L.key[c];
        const double tc = L.ts[c];
        int first = 1;
        for (int q = 0; q < c; ++q)
            if (L.key[q] == kc && L.ts[q] == tc) { first = 0; break; }
        L.perm[c] = first;
    }
    wave_sync();
    // The BFS frontier above needed the un-merged weights; from here on only
    // leaders matter.  A leader's value is the left-to-right sum of its
    // occurrences (tppr_dict[state] = tppr_dict[state] + weight, :222-225).
    // Written in place: a leader only reads its own weight and those of later
    // NON-leader occurrences, which no one writes.
    for (int c = lane; c < n_cand; c += WAVE) {
        if (L.perm[c]) {
            double v = L.w[c];
            const u64 kc = L.key[c];
            const double tc = L.ts[c];
            for (int q = c + 1; q < n_cand; ++q)
                if (L.key[q] == kc && L.ts[q] == tc) v = v + L.w[q];
            L.w[c] = v;
        }
    }
    wave_sync();
    // compact leaders in order (dictionary insertion order)
    int nd = 0;
    for (int c0 = 0; c0 < n_cand; c0 += WAVE) {
        const int c = c0 + lane;
        const bool lead = c < n_cand && L.perm[c] != 0;
        const u64 kc = lead ? L.key[c] : 0;
        const double tc = lead ? L.ts[c] : 0.0;
        const double wc = lead ? L.w[c] : 0.0;
        const u64 bm = __ballot(lead);
        wave_sync();   // all reads of this chunk done before it may be overwritten
        if (lead) {
            const int pos = nd + __popcll(bm & lanemask_lt());   // pos <= c: never clobbers unread chunks
            L.key[pos] = kc; L.ts[pos] = tc; L.w[pos] = wc;
        }
        nd += __popcll(bm);
        wave_sync();
    }

    // ---- select and emit (:240-276) ----
    const long long ob = qi * k;
    if (nd <= k) {
        if (lane < k) {
            const bool a = lane < nd;
            out_nodes[ob + lane] = a ? (int)(unsigned)(L.key[lane] & 0xffffffffull) : 0;
            out_eidx[ob + lane] = a ? (int)(unsigned)(L.key[lane] >> 32) : 0;
            out_w[ob + lane] = a ? (float)L.w[lane] : 0.f;
            const float tsf = a ? (float)L.ts[lane] : 0.f;
            out_dt[ob + lane] = (float)(qt - (double)tsf);
        }
        return;
    }
    topk_select_wave(L.w, nd, k, L.sel, *L.sort, L.perm, L.stk);
    if (lane < k) {
        const int c = L.sel[lane];
        out_nodes[ob + lane] = (int)(unsigned)(L.key[c] & 0xffffffffull);
        out_eidx[ob + lane] = (int)(unsigned)(L.key[c] >> 32);
        out_w[ob + lane] = (float)L.w[c];
        out_dt[ob + lane] = (float)(qt - (double)(float)L.ts[c]);
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
    long long cap = 0, lvl = 1, front = 1;
    for (int d = 0; d < depth; ++d) { front = lvl; lvl *= width; cap += lvl; if (cap > MAX_CAND) break; }
    if (k > ZT_MAX_K || cap > MAX_CAND || front > MAX_FRONT) {
        set_error("zt_pruned_topk: k=%d width=%d depth=%d exceeds the LDS-resident limits "
                  "(k<=%d, sum width^d<=%d)", k, width, depth, ZT_MAX_K, MAX_CAND);
        return ZT_ERR_UNSUPPORTED;
    }
    const int cap_c = (int)cap, cap_f = (int)front;
    const size_t lds = prune_lds_bytes(cap_c, cap_f) * PR_WAVES;
    static size_t attr_lds = 0;
    if (lds > 48 * 1024 && lds > attr_lds) {
        ZT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_pruned_topk),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_lds = lds;
    }
    const int grid = (int)((nq + PR_WAVES - 1) / PR_WAVES);
    ZT_PROF_BEGIN((hipStream_t)stream, P_PRUNE);
    k_pruned_topk<<<grid, WAVE * PR_WAVES, lds, (hipStream_t)stream>>>(
        c->indptr, c->nbr, c->eid, c->ts, c->N, q_nodes_dev, q_ts_dev, nq, width, depth, alpha, beta, k,
        out_nodes_dev, out_eidx_dev, out_dt_dev, out_w_dev, status_dev, cap_c, cap_f);
    ZT_PROF_END((hipStream_t)stream, P_PRUNE);
    ZT_LAUNCH_CHECK();
    return ZT_OK;
}
