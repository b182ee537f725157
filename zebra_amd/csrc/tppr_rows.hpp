// Rows of the streaming T-PPR state and the merges on them (device code of k_stream, tppr_stream.hip): tagged-granule row
// loads / stores, the LDS merge, the register-resident merge and its bitonic front half, waits with time-outs.
#pragma once

#include "numba_sort.hpp"
#include "tppr_state.hpp"

using namespace zt;

namespace {


constexpr int CAP = 128;          // candidates per merge: 2k+1 <= 127
// The register-resident merge (and with it the hub chains) keeps s2's entries in lanes 32 + j, the new key behind them,
// and uses lane 63 as the lane nobody reads in its cross-lane pushes: 32 + k <= 62.  (k = 31 put the new key of a full
// partner row INTO lane 63 -- found by tests/soak_tppr.py, never by the fixed-size tests: k = 5, 20, 40, 63.)
constexpr int REG_K_MAX = 30;
constexpr int WAVES_PER_WG = 8;
constexpr long long WAIT_TICKS = 400000000ll;  // 4 s of the 100 MHz wall clock: bound on any dependency wait

// -DZT_CRIT (diagnostic build, tools/crit_profile.py): core-clock readings at four points of a hub hop, kept in
// registers and written out at the end of the hop -- the chain itself is not disturbed by stores
#ifdef ZT_CRIT
__device__ long long g_crit[8200 * 16];    // per hub edge (model 0): see tools/crit_profile.py
#define CRIT(j) do { crit_t[j] = (long long)__builtin_readcyclecounter(); } while (0)
#define CRITP(j) do { if (crit_p) crit_p[j] = (long long)__builtin_readcyclecounter(); } while (0)
#define CRIT_ARG , long long *crit_p = nullptr
#define CRIT_PASS , crit_t
#else
#define CRIT(j) do { } while (0)
#define CRITP(j) do { } while (0)
#define CRIT_ARG
#define CRIT_PASS
#endif

#ifdef ZT_STAMP
__device__ int g_paths[8];
__device__ long long g_stamps[8192 * 4];
__device__ long long g_stamps2[8192 * 8];
#define STAMP2(slot) do { if (lane_id() == 0 && g_stamp_i >= 0 && g_stamp_i < 8192) g_stamps2[g_stamp_i * 8 + (slot)] = (long long)wall_clock64(); } while (0)   // diagnostic build only: per task t_deq, t_rows, t_x1, t_end (100 MHz ticks)
#define STAMP(slot) do { if (lane == 0 && mo == 0 && i < 8192) g_stamps[i * 4 + (slot)] = (long long)wall_clock64(); } while (0)
// hub hops, per (chain, position) of model 0: 0 entered, 1 partner's row in hand and prepared, 2 turn arrived, 3 end,
// 4 publication; partner halves, per edge: 4 entered, 5 partner's and negative's rows in hand, 6 hub's version in hand,
// 7 partner's row stored
__device__ long long g_hopst[16 * 2048 * 8];
__device__ long long g_stamps3[8192 * 8];
#define HSTAMP(slot) do { if (lane == 0 && mo == 0 && chain_idx >= 0 && chain_idx < 16 && tpos < 2048) g_hopst[(chain_idx * 2048 + tpos) * 8 + (slot)] = (long long)wall_clock64(); } while (0)
#define STAMP3(slot) do { if (lane == 0 && mo == 0 && i < 8192) g_stamps3[i * 8 + (slot)] = (long long)wall_clock64(); } while (0)
#else
#define STAMP(slot) do { } while (0)
#define STAMP3(slot) do { } while (0)
#define HSTAMP(slot) do { } while (0)
#define STAMP2(slot) do { } while (0)
#endif

#ifdef ZT_WAITLOG
// diagnostic build only: per (model, edge) task [state, wg*4+wave, wait kind, target, expect, chain pos, by_mail, clock]
__device__ int g_wl[2 * MAX_CHUNK * 8];
#define WL(f, v) do { if (mo < 2 && lane == __builtin_ctzll(__ballot(1))) __hip_atomic_store(&g_wl[(mo * MAX_CHUNK + i) * 8 + (f)], (int)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (0)
#else
#define WL(f, v) do { } while (0)
#endif

constexpr int HTAB = 1024;        // slots of a wave's key-hash table (all -1 between uses)

struct WaveLds {
    u64 key[CAP];
    double ts[CAP];
    double w[CAP];
    int sel[64];
    SortLds sort;
    int htab[HTAB];
};

__device__ __forceinline__ int key_hash(u64 key)
{
    return (int)((((unsigned)key * 0x9E3779B1u) ^ ((unsigned)(key >> 32) * 0x85EBCA77u)) >> 22);
}

// a second, independent slot function: a partner row whose keys collide under key_hash rarely collides under this one
__device__ __forceinline__ int key_hash2(u64 key)
{
    return (int)((((unsigned)key * 0x85EBCA77u) ^ ((unsigned)(key >> 32) * 0xC2B2AE3Du)) >> 22);
}

// ... and a third (chain hops: both of the others clash for ~2 % of the partner rows)
__device__ __forceinline__ int key_hash3(u64 key)
{
    return (int)((((unsigned)key * 0x27D4EB2Fu) ^ ((unsigned)(key >> 32) * 0x165667B1u)) >> 22);
}
// by the code chain_hop keeps in pre_hash: 1, 3, 5 = the table was filled with slot function 1, 2, 3
__device__ __forceinline__ int key_hash_by(u64 key, int code)
{
    return code == 5 ? key_hash3(key) : (code == 3 ? key_hash2(key) : key_hash(key));
}
// ... and the same with the slot function's two multipliers handed over (key_hash_muls): inside a hub hop's critical
// section the choice by code was a chain of exec-masked branches, ~20 instructions per probe (round 6: an instruction on the
// chain costs 0.09 % of the kernel)
__device__ __forceinline__ void key_hash_muls(int code, unsigned &m0, unsigned &m1)
{
    m0 = code == 5 ? 0x27D4EB2Fu : (code == 3 ? 0x85EBCA77u : 0x9E3779B1u);
    m1 = code == 5 ? 0x165667B1u : (code == 3 ? 0xC2B2AE3Du : 0x85EBCA77u);
}
__device__ __forceinline__ int key_hash_m(u64 key, unsigned m0, unsigned m1)
{
    return (int)((((unsigned)key * m0) ^ ((unsigned)(key >> 32) * m1)) >> 22);
}

// ------------------------------------------------------------- row access ----
struct Row {
    u64 key;
    double ts, w;   // this lane's entry (valid for lane < len)
    int len;
    double norm;
};

// Loads one row.  expect != 0: every granule must carry that tag; returns the
// first mismatching tag seen (or `expect` when the row is complete).
__device__ __forceinline__ unsigned load_row_at(const u64 *base, int k, int lane, unsigned expect, Row &r);
__device__ __forceinline__ unsigned load_row(const zt_tppr &h, int m, long long x, int lane, unsigned expect, Row &r)
{
    return load_row_at(h.rows + ((long long)m * h.N + x) * h.rg, h.k, lane, expect, r);
}
// The granules of a row as they come from memory: load_row_issue starts the loads, row_from_raw (which waits for them)
// unpacks and checks the tags -- apart, so that a chain wave can have the partner's row of its NEXT hop on its way
// while it finishes the current one.
struct RawRow {
    u64 g0, g1, g2, g3, g4, g5, gh;
};
__device__ __forceinline__ void load_row_issue(const u64 *base, int k, int lane, RawRow &q)
{
    q.g0 = q.g1 = q.g2 = q.g3 = q.g4 = q.g5 = q.gh = 0;
    if (lane < 3) q.gh = ld_agent(base + lane);
    if (lane < k) {
        const u64 *e = base + HDR + lane;
        q.g0 = ld_agent(e);
        q.g1 = ld_agent(e + k);
        q.g2 = ld_agent(e + 2 * k);
        q.g3 = ld_agent(e + 3 * k);
        q.g4 = ld_agent(e + 4 * k);
        q.g5 = ld_agent(e + 5 * k);
    }
}
__device__ __forceinline__ unsigned row_from_raw(const RawRow &q, int k, int lane, unsigned expect, Row &r);
__device__ __forceinline__ unsigned load_row_at(const u64 *base, int k, int lane, unsigned expect, Row &r)
{
    RawRow q;
    load_row_issue(base, k, lane, q);
    return row_from_raw(q, k, lane, expect, r);
}
__device__ __forceinline__ unsigned row_from_raw(const RawRow &q, int k, int lane, unsigned expect, Row &r)
{
    const u64 g0 = q.g0, g1 = q.g1, g2 = q.g2, g3 = q.g3, g4 = q.g4, g5 = q.g5, gh = q.gh;
    const unsigned h0 = (unsigned)__shfl((unsigned)gh, 0), h1 = (unsigned)__shfl((unsigned)gh, 1),
                   h2 = (unsigned)__shfl((unsigned)gh, 2);
    r.len = (int)h0;
    r.norm = __longlong_as_double((long long)(((u64)h2 << 32) | h1));
    r.key = ((u64)(unsigned)g1 << 32) | (unsigned)g0;
    r.ts = __longlong_as_double((long long)(((u64)(unsigned)g3 << 32) | (unsigned)g2));
    r.w = __longlong_as_double((long long)(((u64)(unsigned)g5 << 32) | (unsigned)g4));
    if (expect == 0) return 0;
    unsigned bad = expect;
    if (lane < 3 && (unsigned)(gh >> 32) != expect) bad = (unsigned)(gh >> 32);
    if (lane < k) {
        const unsigned t0 = (unsigned)(g0 >> 32), t1 = (unsigned)(g1 >> 32), t2 = (unsigned)(g2 >> 32),
                       t3 = (unsigned)(g3 >> 32), t4 = (unsigned)(g4 >> 32), t5 = (unsigned)(g5 >> 32);
        if (t0 != expect) bad = t0;
        if (t1 != expect) bad = t1;
        if (t2 != expect) bad = t2;
        if (t3 != expect) bad = t3;
        if (t4 != expect) bad = t4;
        if (t5 != expect) bad = t5;
    }
    const u64 bm = __ballot(bad != expect);
    if (bm == 0ull) return expect;
    return (unsigned)__shfl(bad, __ffsll((long long)bm) - 1);
}

// extract_streaming_tppr (utils/util.py:447-469)
__device__ __forceinline__ void emit_row(const Row &r, int k, int lane, double tnow, int *on, int *oe, float *od,
                                         float *ow)
{
    // (write-through stores, as the rows': a consumer on another XCD may be let at a batch's rows while this launch is still
    //  running -- StreamArgs::member_done -- and must find them in memory, not in this XCD's L2)
    if (lane >= k) return;
    if (r.len == 0) { st_agent(on + lane, 0); st_agent(oe + lane, 0); st_agent(od + lane, 0.f); st_agent(ow + lane, 0.f); return; }
    const bool a = lane < r.len;
    st_agent(on + lane, a ? (int)(unsigned)(r.key & 0xffffffffull) : 0);
    st_agent(oe + lane, a ? (int)(unsigned)(r.key >> 32) : 0);
    st_agent(ow + lane, a ? (float)r.w : 0.f);
    const float tsf = a ? (float)r.ts : 0.f;      // tmp_timestamps is float32
    st_agent(od + lane, (float)(tnow - (double)tsf));        // f64 - f32 -> f64 -> stored f32
}

// One (s1, s2) pair of the update block (utils/util.py:509-564).  Returns the
// new length of s1's dictionary; lane j < length holds entry j in (ok, ot, ow).
__device__ __forceinline__ u64 readlane_u64(u64 x, int src /* wave-uniform */)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)x, src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(x >> 32), src);
    return ((u64)hi << 32) | lo;
}

__device__ inline int merge_pair(WaveLds &L, int lane, int k, double alpha, double beta, const Row &r1,
                                 const Row &r2, u64 newkey, double newts, u64 &ok, double &ot, double &ow,
                                 int pre = 0, int g_stamp_i = -1)
{
    STAMP2(0);
    // A lone wave is bound by dependent-instruction latency, so this routine is written for the
    // length of its dependency chain: s1's entries live in lanes [0,n1), s2's in lanes [0,len2),
    // key matches are found for all pairs at once through LDS, and the merged candidate list is
    // written to LDS once (the top-k prune permutes it).
    int n1;
    double scale_s1 = 0.0, scale_s2;
    if (r1.norm == 0.0) {                       // :514-519
        n1 = 0;
        scale_s2 = 1.0 - alpha;
    } else {                                    // :520-527
        n1 = r1.len;
        const double new_norm = r1.norm * beta + beta;
        scale_s1 = r1.norm / new_norm * beta;
        scale_s2 = beta / new_norm * (1.0 - alpha);
    }
    const bool in1 = lane < n1;
    double w1 = r1.w * scale_s1;                // t_s1_PPR[key] = value * scale_s1
    const int len2 = (r2.norm != 0.0) ? r2.len : 0;   // :530-538
    const bool in2 = lane < len2;
    const double add = r2.w * scale_s2;
    bool matched2 = false;                      // this lane's s2 entry met its key in s1
    bool matched_done = false;
    if (n1 > 0 && len2 > 0) {
        // Key matching, fast path: s2's entries enter a hash table in LDS (slot = hash of the key, value =
        // lane); if no two of them share a slot, an s1 entry can only match the entry in ITS slot, which
        // it fetches from the lane holding it and compares in full.  Keys are unique inside a
        // dictionary, so an entry has at most one partner.  Slot collisions (~1 call in 6 for 20
        // entries in 1024 slots) take the all-pairs path below.
        // (pre != 0: s2 was entered into the table by the caller while it waited for r1 -- hub chains)
        int *T = L.htab;
        const int h1 = key_hash(r1.key), h2 = key_hash(r2.key);
        bool clash = pre == 2;
        if (pre == 0) {
            if (in2) T[h2] = lane;
            L.sort.r[lane] = 0;
            wave_sync();
            const int back = in2 ? T[h2] : lane;
            clash = __ballot(in2 && back != lane) != 0ull;
            if (clash && in2 && back == lane) T[h2] = -1;       // the slot's last writer clears it
        }
        const int cand = (in1 && !clash) ? T[h1] : -1;
        if (!clash) {
            const int src = cand >= 0 ? cand : 0;
            const u64 kj = __shfl(r2.key, src);
            const double tj = __shfl(r2.ts, src), aj = __shfl(add, src);
            const bool hit = in1 && cand >= 0 && kj == r1.key && tj == r1.ts;
            if (hit) { w1 = w1 + aj; L.sort.r[cand] = 1; }      // t_s1_PPR[key] += value * scale_s2
            if (in2) T[h2] = -1;
            wave_sync();
            matched2 = in2 && L.sort.r[lane] != 0;
            matched_done = true;
        }
        wave_sync();
    }
    if (pre == 1 && !(n1 > 0 && len2 > 0) && lane < r2.len) L.htab[key_hash(r2.key)] = -1;   // (cannot happen on a chain)
    if (n1 > 0 && len2 > 0 && !matched_done) {
        // Key matching through LDS, all pairs at once: the rows are staged (s1 in slots [0,64), s2
        // in [64,128)), lane (c, i) compares s1's entry i with every S-th entry of s2 starting at c.
        // Keys are unique inside a dictionary, so an entry has at most one partner.  Four dependent
        // LDS round trips instead of one broadcast + ballot per entry of the shorter row.
        int *m1 = L.sel, *m2 = L.sort.r;
        if (in1) { L.key[lane] = r1.key; L.ts[lane] = r1.ts; }
        if (in2) { L.key[WAVE + lane] = r2.key; L.ts[WAVE + lane] = r2.ts; L.w[WAVE + lane] = add; }
        m1[lane] = -1;
        m2[lane] = 0;
        wave_sync();
        // S lanes share one s1 entry: lane = c * n1 + i probes s2's entries c, c + S, c + 2S, ...
        const int S = n1 <= 16 ? 4 : (n1 <= 21 ? 3 : (n1 <= 32 ? 2 : 1));
        const int c = (lane >= n1 ? 1 : 0) + (lane >= 2 * n1 ? 1 : 0) + (lane >= 3 * n1 ? 1 : 0);
        const int i = lane - c * n1;
        if (c < S && i < n1) {
            const u64 ki = L.key[i];
            const double ti = L.ts[i];
            int jm = -1;
            for (int j0 = c; j0 < len2; j0 += 8 * S) {            // eight probes in flight
                u64 kj[8];
                double tj[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int j = j0 + t * S, jj = j < len2 ? j : len2 - 1;
                    kj[t] = L.key[WAVE + jj]; tj[t] = L.ts[WAVE + jj];
                }
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int j = j0 + t * S;
                    if (j < len2 && kj[t] == ki && tj[t] == ti) jm = j;
                }
            }
            if (jm >= 0) { m1[i] = jm; m2[jm] = 1; L.sort.v[i] = L.w[WAVE + jm]; }
        }
        wave_sync();
        if (in1 && m1[lane] >= 0) w1 = w1 + L.sort.v[lane];       // t_s1_PPR[key] += value * scale_s2
        matched2 = in2 && m2[lane] != 0;
        wave_sync();                                              // the staging area is reused below
    }
    const bool un2 = in2 && !matched2;          // appended after s1's entries, in s2's order
    const u64 um = __ballot(un2);
    const int pos2 = n1 + __popcll(um & lanemask_lt());
    int n = n1 + __popcll(um);
    // new key (edge_idx, s2, ts): overwrite if present, else append last (:531 / :540-541)
    const double v = (alpha != 0.0) ? scale_s2 * alpha : scale_s2;
    const bool h1 = in1 && r1.key == newkey && r1.ts == newts;
    const bool h2 = un2 && r2.key == newkey && r2.ts == newts;
    const bool present = __ballot(h1 || h2) != 0ull;
    if (h1) w1 = v;
    const double w2 = h2 ? v : add;
    STAMP2(1);
    if (in1) { L.key[lane] = r1.key; L.ts[lane] = r1.ts; L.w[lane] = w1; }
    if (un2) { L.key[pos2] = r2.key; L.ts[pos2] = r2.ts; L.w[pos2] = w2; }
    if (!present) {
        if (lane == 0) { L.key[n] = newkey; L.ts[n] = newts; L.w[n] = v; }
        ++n;
    }
    wave_sync();
    STAMP2(2);
    STAMP2(3);
    if (n <= k) {                               // :549-551
        if (lane < n) { ok = L.key[lane]; ot = L.ts[lane]; ow = L.w[lane]; }
        wave_sync();
        return n;
    }
    const int tk_path = topk_select_wave(L.w, n, k, L.sel, L.sort, L.sort.r, L.sort.stk);   // :553-559
#ifdef ZT_STAMP
    if (lane == 0) atomicAdd(&g_paths[tk_path & 7], 1);
#endif
    (void)tk_path;
    STAMP2(4);
    if (lane < k) {
        const int c = L.sel[lane];
        ok = L.key[c]; ot = L.ts[c]; ow = L.w[c];
    }
    wave_sync();
    STAMP2(5);
    return k;
}

// A merged dictionary as the register-resident merge leaves it: this lane's candidate (if any) and the slot
// [0, n_new) it takes in the new dictionary (-1: dropped / no candidate).  The new row is never brought into
// lane order: mailbox and row stores scatter straight from the candidate lanes.
struct Cand {
    u64 key;
    double ts, w;
    int slot;
};

// What the FRONT half of a pair update knows (merge_front): the candidates and, if a prune is needed, their
// ranks -- everything that does not depend on the ORDER of s1's entries.  merge_order finishes the job once
// the dictionary position of every s1 entry is known.  On a hub chain the two halves are separated by the
// arrival of the previous hop's order (process_edge).
enum { FR_NOPRUNE = 0, FR_RANKS = 1, FR_TIES = 2, FR_STRADDLE = 3, FR_NAN = 4 };
struct Front {
    u64 key;
    double ts, w;      // this lane's candidate (valid if live)
    u64 live;          // uniform: lanes holding a candidate (s1's entries in [0, n1), the rest from lane 32)
    int n, n1;         // uniform: candidates, s1's entries among them
    int pos_tail;      // lanes >= 32: dictionary position of the candidate (s1's entries precede them)
    int lt;            // strictly smaller candidates (n > k)
    int mode;          // FR_*: no prune / ranks decide / ties decide the order / ... and the kept set / NaN
    bool keep;         // FR_RANKS, FR_TIES: this lane's candidate is kept
    unsigned claimed;  // uniform: bit r = some candidate has exactly drop + r smaller ones (rank_pass)
    u64 touched;       // uniform: s1 lanes whose weight a key match or the new key changed
    // merge_front_fast only (fast): the candidates ascending by weight occupy sorted positions 0 .. n-1
    bool fast;
    int sp;            // sorted position of this lane's candidate
    u64 S;             // uniform: positions where a run of equal weights starts
};

// The same pair update with the candidate list held in REGISTERS, for k <= REG_K_MAX = 30: s1's entries stay in lanes
// [0, n1), s2's entries move to lanes 32 + j with ONE v_permlane32_swap per register (no LDS), the new key
// sits behind them.  The top-k works on these register values (topk_reg: ranks by lane broadcasts, tie
// test by a DPP OR-reduction, quicksort replay on the compacted ranks only when ties decide).  Compared
// with merge_pair the candidate list, the selection vector and the gathered result never touch LDS.
// The scale factors of an update depend on the row's norm only, and along a hub chain the norm follows
// norm <- norm * beta + beta from hop to hop: a wave works them out (two float64 divisions) for the norm it
// EXPECTS while it waits for the mailbox; merge_front takes them if the norm that arrives is that one.
struct PreScale {
    double norm, scale_s1, scale_s2;
    double norm_next;  // norm * beta + beta
    bool valid;
};

__device__ inline void merge_front(WaveLds &L, int lane, int k, double alpha, double beta, const Row &r1,
                                   const Row &r2, u64 newkey, double newts, Front &F, int pre = 0,
                                   int g_stamp_i = -1, const PreScale *ps = nullptr CRIT_ARG)
{
    STAMP2(0);
    CRITP(4);
    int n1;
    double scale_s1 = 0.0, scale_s2;
    if (r1.norm == 0.0) {                       // :514-519
        n1 = 0;
        scale_s2 = 1.0 - alpha;
    } else if (ps != nullptr && ps->valid && ps->norm == r1.norm) {
        n1 = __builtin_amdgcn_readfirstlane(r1.len);
        scale_s1 = ps->scale_s1;                // the same expressions on the same norm, evaluated ahead
        scale_s2 = ps->scale_s2;
    } else {                                    // :520-527
        n1 = __builtin_amdgcn_readfirstlane(r1.len);        // row headers are wave-uniform: say so
        const double new_norm = r1.norm * beta + beta;
        scale_s1 = r1.norm / new_norm * beta;
        scale_s2 = beta / new_norm * (1.0 - alpha);
    }
    const bool in1 = lane < n1;
    double w1 = r1.w * scale_s1;                // t_s1_PPR[key] = value * scale_s1
    const int len2 = __builtin_amdgcn_readfirstlane((r2.norm != 0.0) ? r2.len : 0);   // :530-538
    const bool in2 = lane < len2;
    const double add = r2.w * scale_s2;
    u64 m2mask = 0ull;                          // lanes of s2 whose key is already in s1
    u64 touched = 0ull;                         // lanes of s1 that a match (or the new key, below) lands on
    if (n1 > 0 && len2 > 0) {
        // hash matching as in merge_pair; which of s2's lanes were hit is collected from the (rare)
        // hitting lanes by scalar reads instead of a flag array in LDS
        int *T = L.htab;
        const int h1 = key_hash(r1.key), h2 = key_hash(r2.key);
        bool clash = pre == 2;
        if (pre == 0) {
            if (in2) T[h2] = lane;
            wave_sync();
            const int back = in2 ? T[h2] : lane;
            clash = __ballot(in2 && back != lane) != 0ull;
            if (clash && in2 && back == lane) T[h2] = -1;       // the slot's last writer clears it
        }
        if (!clash) {
            const int cand = in1 ? T[h1] : -1;
            if (in2) T[h2] = -1;
            if (__ballot(cand >= 0) != 0ull) {                  // mostly no slot is even occupied
                const int src = cand >= 0 ? cand : 0;
                const u64 kj = __shfl(r2.key, src);
                const double tj = __shfl(r2.ts, src), aj = __shfl(add, src);
                const bool hit = in1 && cand >= 0 && kj == r1.key && tj == r1.ts;
                if (hit) w1 = w1 + aj;                          // t_s1_PPR[key] += value * scale_s2
                u64 hm = __ballot(hit);
                touched = hm;
                while (hm) {
                    const int l = __ffsll((long long)hm) - 1;
                    hm &= hm - 1ull;
                    m2mask |= 1ull << __builtin_amdgcn_readlane(cand, l);
                }
            }
        } else {
            // slot collision: all pairs through LDS (merge_pair's fallback), results back into registers
            int *m1 = L.sel, *m2 = L.sort.r;
            wave_sync();
            if (in1) { L.key[lane] = r1.key; L.ts[lane] = r1.ts; }
            if (in2) { L.key[WAVE + lane] = r2.key; L.ts[WAVE + lane] = r2.ts; L.w[WAVE + lane] = add; }
            m1[lane] = -1;
            m2[lane] = 0;
            wave_sync();
            const int S = n1 <= 16 ? 4 : (n1 <= 21 ? 3 : (n1 <= 32 ? 2 : 1));
            const int c = (lane >= n1 ? 1 : 0) + (lane >= 2 * n1 ? 1 : 0) + (lane >= 3 * n1 ? 1 : 0);
            const int i = lane - c * n1;
            if (c < S && i < n1) {
                const u64 ki = L.key[i];
                const double ti = L.ts[i];
                int jm = -1;
                for (int j0 = c; j0 < len2; j0 += 8 * S) {
                    u64 kj[8];
                    double tj[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const int j = j0 + t * S, jj = j < len2 ? j : len2 - 1;
                        kj[t] = L.key[WAVE + jj]; tj[t] = L.ts[WAVE + jj];
                    }
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const int j = j0 + t * S;
                        if (j < len2 && kj[t] == ki && tj[t] == ti) jm = j;
                    }
                }
                if (jm >= 0) { m1[i] = jm; m2[jm] = 1; L.sort.v[i] = L.w[WAVE + jm]; }
            }
            wave_sync();
            if (in1 && m1[lane] >= 0) w1 = w1 + L.sort.v[lane];
            touched = __ballot(in1 && m1[lane] >= 0);
            m2mask = __ballot(in2 && m2[lane] != 0);
            wave_sync();
        }
    }
    if (pre == 1 && !(n1 > 0 && len2 > 0) && lane < r2.len) L.htab[key_hash(r2.key)] = -1;   // (cannot happen on a chain)
    const bool un2 = in2 && ((m2mask >> lane) & 1ull) == 0ull;    // appended after s1's entries, in s2's order
    const u64 um = __ballot(un2);                                  // bits < 32
    int n = n1 + __popcll(um);
    // new key (edge_idx, s2, ts): overwrite if present, else append last (:531 / :540-541)
    const double v = (alpha != 0.0) ? scale_s2 * alpha : scale_s2;
    const bool h1 = in1 && r1.key == newkey && r1.ts == newts;
    const bool h2 = un2 && r2.key == newkey && r2.ts == newts;
    const bool present = __ballot(h1 || h2) != 0ull;
    if (h1) w1 = v;
    touched |= __ballot(h1);
    const double w2 = h2 ? v : add;
    STAMP2(1);
    CRITP(5);
    // candidates: lanes [0, n1) s1's entries, lanes 32 + j s2's unmatched entries, lane 32 + len2 the new key.
    // v_permlane32_swap(a, b) exchanges a[32..63] with b[0..31]: the first result is [a's low half | b's low half].
    u64 ck;
    double ct, cw;
    {
        const unsigned a0 = (unsigned)r1.key, a1 = (unsigned)(r1.key >> 32), b0 = (unsigned)r2.key, b1 = (unsigned)(r2.key >> 32);
        const u64 ta = (u64)__double_as_longlong(r1.ts), tb = (u64)__double_as_longlong(r2.ts);
        const u64 wa = (u64)__double_as_longlong(w1), wb = (u64)__double_as_longlong(w2);
#define ZT_SWAP(x, y) ((unsigned)__builtin_amdgcn_permlane32_swap((x), (y), false, false)[0])
        ck = ((u64)ZT_SWAP(a1, b1) << 32) | ZT_SWAP(a0, b0);
        ct = __longlong_as_double((long long)(((u64)ZT_SWAP((unsigned)(ta >> 32), (unsigned)(tb >> 32)) << 32) | ZT_SWAP((unsigned)ta, (unsigned)tb)));
        cw = __longlong_as_double((long long)(((u64)ZT_SWAP((unsigned)(wa >> 32), (unsigned)(wb >> 32)) << 32) | ZT_SWAP((unsigned)wa, (unsigned)wb)));
#undef ZT_SWAP
    }
    u64 live = (n1 > 0 ? ((1ull << n1) - 1ull) : 0ull) | (um << 32);
    int pos = lane < 32 ? lane : n1 + __popcll((um << 32) & lanemask_lt());     // place in the reference's dictionary order
    if (!present) {
        const int nl = 32 + len2;                                               // <= 62 (len2 <= k <= 30)
        if (lane == nl) { ck = newkey; ct = newts; cw = v; pos = n; }
        live |= 1ull << nl;
        ++n;
    }
    live = ((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(live >> 32)) << 32) |
           (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)live);
    n = __builtin_amdgcn_readfirstlane(n);
    F.key = ck; F.ts = ct; F.w = cw;
    F.live = live; F.n = n; F.n1 = n1; F.pos_tail = pos;
    F.lt = 0; F.keep = false; F.touched = touched; F.claimed = 0u;
    F.fast = false; F.sp = -1; F.S = 0ull;
    STAMP2(2);
    STAMP2(3);
    const bool mine = (live >> lane) & 1ull;
    if (n <= k) { F.mode = FR_NOPRUNE; return; }          // :549-551
    if (__ballot(mine && cw != cw) != 0ull) { F.mode = FR_NAN; return; }
    CRITP(6);
    F.mode = rank_pass(cw, live, n, k, &F.lt, &F.keep, &F.claimed);    // :553-559 (first half)
    CRITP(7);
#ifdef ZT_STAMP
    if (lane == 0) atomicAdd(&g_paths[F.mode == FR_RANKS ? 0 : (F.mode == FR_TIES ? 4 : 5)], 1);
#endif
    STAMP2(4);
}

// ---------------------------------------------------------------------------------------------------------
// The front half on sorted inputs (hub chains): ranks from a bitonic MERGE instead of 48 lane broadcasts.
// Along a chain the hub's row arrives in ascending order of weight (the arrangement the previous hop published)
// and scaling by scale_s1 keeps that order.  The partner's side -- its scaled entries and the new key -- is
// known while the wave still WAITS for the hub's row: it is sorted there (prepare_b, off the chain), descending,
// into lanes 32..63 behind +inf padding, so that lanes 0..63 form a bitonic sequence once the hub's entries sit in
// lanes 0..n1-1.  Six compare-exchange stages (lane distance 32, 16, 8, 4, 2, 1: permlane swaps and DPP moves,
// no LDS) then leave the 2k+1 candidates ascending in lanes 0..n-1, each with the lane its candidate lives in; a
// run of equal weights starts where a lane differs from its left neighbour, the number of strictly smaller
// candidates of every member is the lane of that start, one ds_permute carries it home, and "do ties reach the
// kept ranks / does a run straddle the cut" are bit tests on the mask of run starts.  The results (lt, keep,
// claimed, mode) are those of rank_pass, bit for bit.
// Preconditions, checked here (false: the caller takes merge_front): the scale factors were predicted for the norm
// that arrived, no key of the partner's row is in the hub's row already (hash probe, verified), the new key is not in
// the hub's row, no NaN.
// ---------------------------------------------------------------------------------------------------------
struct PreB {
    bool ok;           // uniform: the fast path may be tried
    int len2, nb;      // uniform: partner entries, candidates of the partner's side (entries + the new key unless present)
    u64 cb_key;        // lanes >= 32: payload of the partner-side candidate living in this lane (entry j in lane 32 + j,
    double cb_ts, cb_w;   //           the new key in lane 32 + len2), as merge_front lays them out
    double sw;         // lanes >= 32: the same weights sorted DESCENDING behind +inf padding
    int sid;           // lanes >= 32: the lane the candidate of that sorted position lives in (padding: an unused lane)
    int h2;            // lanes < len2: hash slot of the partner's entry (to clear the table)
};

__device__ inline void prepare_b(int lane, int k, double alpha, const Row &r2, u64 newkey, double newts,
                                 const PreScale &ps, PreB &B, int h2slot)
{
    B.ok = false;
    if (!ps.valid || k > REG_K_MAX) return;
    const int len2 = __builtin_amdgcn_readfirstlane((r2.norm != 0.0) ? r2.len : 0);
    const bool in2 = lane < len2;
    const double v = (alpha != 0.0) ? ps.scale_s2 * alpha : ps.scale_s2;           // :531 / :540-541
    const bool h2 = in2 && r2.key == newkey && r2.ts == newts;
    const bool present2 = __ballot(h2) != 0ull;
    const double w2 = h2 ? v : r2.w * ps.scale_s2;                                  // value * scale_s2 (:530-538)
    const int nb = len2 + (present2 ? 0 : 1);
    const bool isnew = !present2 && lane == len2;
    const bool el = lane < nb;                                                      // this lane holds element `lane` of the side
    const double bw = isnew ? v : w2;
    if (__ballot(el && bw != bw) != 0ull) return;                                   // NaN: the general path
    // descending order, equal weights by element number: rb = elements that come before mine
    int rb = 0;
    {
        // A pruned row is stored in ascending order of weight (the reference's dictionary order IS its argsort, :553-559) and
        // scaling keeps that order: then an entry's place follows from the runs of equal weights -- the entries behind its run
        // come before it, and so do the earlier members of its run -- without comparing it with every other one.  (21 lane
        // broadcasts and compares, ~130 instructions, on a compute unit whose instruction issue bounds the chain: round 6.)
        // Rows that never were full (or hold the new key already, whose weight is replaced) take the loop.
        const long long bwb = __double_as_longlong(bw);
        const int plo = __builtin_amdgcn_mov_dpp((int)(unsigned)(bwb & 0xffffffffll), 0x138, 0xf, 0xf, true);   // wave_shr:1
        const int phi = __builtin_amdgcn_mov_dpp((int)(bwb >> 32), 0x138, 0xf, 0xf, true);
        const double prev = __longlong_as_double(((long long)phi << 32) | (unsigned)plo);
        const bool inr = lane > 0 && lane < len2;
        const bool sorted_in = __ballot(inr && bw < prev) == 0ull;
        if (sorted_in) {
            const unsigned starts = (unsigned)__ballot(in2 && (lane == 0 || bw != prev));       // len2 <= 30: bits 0 .. len2-1
            const unsigned upto = starts & ((2u << (lane & 31)) - 1u);                            // run starts at or below my entry
            const int rs = 31 - __clz((int)upto);                                                 // start of my run
            const unsigned after = starts & ~((2u << (lane & 31)) - 1u);
            const int re = after ? __ffs((int)after) - 2 : len2 - 1;                              // its last member
            const int ge_new = __popcll(__ballot(in2 && bw >= v));                                // entries that come before the new key
            rb = in2 ? (len2 - 1 - re) + (lane - rs) + ((!present2 && v > bw) ? 1 : 0) : ge_new;
        } else {
            for (int q = 0; q < nb; ++q) {
                const double x = readlane_f64(bw, q);
                rb += (x > bw || (x == bw && q < lane)) ? 1 : 0;
            }
        }
    }
    // sorted lane of my element; lanes without one push to lane 0 (nobody reads the low half of these registers)
    const int dst = el ? 64 - nb + rb : 0;
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    double sw = push_f64(bw, dst);
    int sid = push_i32(32 + lane, dst);
    const bool pad = lane >= 32 && lane < 64 - nb;                                  // sorted lanes nobody pushed to
    sw = pad ? inf : sw;
    sid = pad ? lane + nb : sid;                      // unused home lanes 32 + nb .. 63, one each
    // payload where merge_front puts it: partner entry j -> lane 32 + j, the new key -> lane 32 + len2
    {
        const unsigned b0 = (unsigned)r2.key, b1 = (unsigned)(r2.key >> 32);
        const u64 tb = (u64)__double_as_longlong(r2.ts), wb = (u64)__double_as_longlong(w2);
#define ZT_SWAP(x, y) ((unsigned)__builtin_amdgcn_permlane32_swap((x), (y), false, false)[0])
        B.cb_key = ((u64)ZT_SWAP(0u, b1) << 32) | ZT_SWAP(0u, b0);
        B.cb_ts = __longlong_as_double((long long)(((u64)ZT_SWAP(0u, (unsigned)(tb >> 32)) << 32) | ZT_SWAP(0u, (unsigned)tb)));
        B.cb_w = __longlong_as_double((long long)(((u64)ZT_SWAP(0u, (unsigned)(wb >> 32)) << 32) | ZT_SWAP(0u, (unsigned)wb)));
#undef ZT_SWAP
        if (!present2 && lane == 32 + len2) { B.cb_key = newkey; B.cb_ts = newts; B.cb_w = v; }
    }
    B.sw = sw; B.sid = sid; B.len2 = len2; B.nb = nb;
    B.h2 = h2slot;
    B.ok = true;
}

// One compare-exchange stage of the bitonic merge on (weight, home lane): of the lanes i and i ^ D the lower keeps the
// smaller weight, the upper the larger; equal weights stay where they are.  The vector unit issues one instruction
// every four cycles or so for the wave that holds the chain, so the stage is written for instruction count:
//   D = 32, 16: v_permlane32/16_swap of a register with its own copy leaves BOTH members of every pair in both lanes
//               (X = the lower member, Y = the upper one): one compare, the mask flipped for the upper lanes on the
//               scalar unit, three selects;
//   D < 16    : the partner's value comes by DPP (row_ror:8, bank-masked row_shl/shr:4, quad_perm); ONE compare: a
//               pair exchanges iff the lower lane sees a smaller partner, and the upper lane's decision is the same
//               bit, shifted by D on the scalar unit.
// __builtin_amdgcn_inverse_ballot_w64 turns the uniform mask into a lane predicate without an instruction.
template <int D>
__device__ __forceinline__ int dpp_xor(int v)
{
    if (D == 8) return __builtin_amdgcn_mov_dpp(v, 0x128, 0xf, 0xf, true);           // row_ror:8
    if (D == 2) return __builtin_amdgcn_mov_dpp(v, 0x4e, 0xf, 0xf, true);            // quad_perm [2,3,0,1]
    if (D == 1) return __builtin_amdgcn_mov_dpp(v, 0xb1, 0xf, 0xf, true);            // quad_perm [1,0,3,2]
    const int t = __builtin_amdgcn_update_dpp(v, v, 0x104, 0xf, 0x5, false);         // row_shl:4 into banks 0, 2 (lane i <- i + 4)
    return __builtin_amdgcn_update_dpp(t, v, 0x114, 0xf, 0xa, false);                // row_shr:4 into banks 1, 3 (lane i <- i - 4)
}

template <int D>
__device__ __forceinline__ void merge_stage(double &w, int &id)
{
    const long long wb = __double_as_longlong(w);
    const int lo = (int)(unsigned)(wb & 0xffffffffll), hi = (int)(wb >> 32);
    if (D >= 16) {
        const auto rl = D == 32 ? __builtin_amdgcn_permlane32_swap(lo, lo, false, false) : __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto rh = D == 32 ? __builtin_amdgcn_permlane32_swap(hi, hi, false, false) : __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        const auto ri = D == 32 ? __builtin_amdgcn_permlane32_swap(id, id, false, false) : __builtin_amdgcn_permlane16_swap(id, id, false, false);
        const double X = __longlong_as_double(((long long)rh[0] << 32) | (unsigned)rl[0]);
        const double Y = __longlong_as_double(((long long)rh[1] << 32) | (unsigned)rl[1]);
        const u64 U = D == 32 ? 0xffffffff00000000ull : 0xffff0000ffff0000ull;       // the upper lanes of the pairs
        const bool take_y = __builtin_amdgcn_inverse_ballot_w64(__ballot(Y < X) ^ U);   // lower: the minimum, upper: the maximum
        w = take_y ? Y : X;
        id = take_y ? ri[1] : ri[0];
    } else {
        const int plo = dpp_xor<D>(lo), phi = dpp_xor<D>(hi), pid = dpp_xor<D>(id);
        const double pw = __longlong_as_double(((long long)phi << 32) | (unsigned)plo);
        const u64 U = D == 8 ? 0xff00ff00ff00ff00ull : D == 4 ? 0xf0f0f0f0f0f0f0f0ull : D == 2 ? 0xccccccccccccccccull : 0xaaaaaaaaaaaaaaaaull;
        const u64 ex = __ballot(pw < w) & ~U;                                        // lower lanes whose pair exchanges
        const bool take = __builtin_amdgcn_inverse_ballot_w64(ex | (ex << D));
        w = take ? pw : w;
        id = take ? pid : id;
    }
}

__device__ inline bool merge_front_fast(WaveLds &L, int lane, int k, const Row &r1, const Row &r2, const PreB &B,
                                        const PreScale &ps, u64 newkey, double newts, bool table_ready, bool hash2, Front &F CRIT_ARG)
{
    CRITP(4);
#ifdef ZT_CRIT
#define FAILC(code) do { if (crit_p) crit_p[11] = (code); } while (0)
#else
#define FAILC(code) do { } while (0)
#endif
    if (!B.ok) { FAILC(1); return false; }
    if (r1.norm == 0.0 || ps.norm != r1.norm) { FAILC(2); return false; }
    const int n1 = __builtin_amdgcn_readfirstlane(r1.len);
    const bool in1 = lane < n1;
    const double w1 = r1.w * ps.scale_s1;                       // t_s1_PPR[key] = value * scale_s1
    const int len2 = B.len2;
    const bool probe = len2 > 0 && n1 > 0;
    if (probe && !table_ready) { FAILC(6); return false; }          // (slot collision among the partner's keys: general path)
    // is a key of the hub's row in the partner's row?  The table read is issued here and looked at AFTER the merge network
    // (verify): its LDS round trip hides behind the network, whose result is thrown away in the rare case of a match
    const int cand = (probe && in1) ? L.htab[hash2 ? key_hash2(r1.key) : key_hash(r1.key)] : -1;
    auto verify = [&]() -> bool {
        bool bad = in1 && ((r1.key == newkey && r1.ts == newts) || w1 != w1);
        if (__ballot(cand >= 0) != 0ull) {                       // an occupied slot: compare the keys in full
            const int src = cand >= 0 ? cand : 0;
            const u64 kj = __shfl(r2.key, src);
            const double tj = __shfl(r2.ts, src);
            bad = bad || (in1 && cand >= 0 && kj == r1.key && tj == r1.ts);          // a real match
        }
        if (__ballot(bad) != 0ull) { FAILC(3); return false; }  // (the table stays as it is: merge_front uses it)
        if (probe && lane < len2) L.htab[B.h2] = -1;             // the table is clean again (merge_front does the same)
        return true;
    };
    CRITP(5);
    const int nb = B.nb, n = n1 + nb;
    F.key = lane < 32 ? r1.key : B.cb_key;
    F.ts = lane < 32 ? r1.ts : B.cb_ts;
    F.w = lane < 32 ? w1 : B.cb_w;
    F.live = (n1 > 0 ? ((1ull << n1) - 1ull) : 0ull) | (((1ull << nb) - 1ull) << 32);
    F.n = n; F.n1 = n1;
    F.pos_tail = lane < 32 ? lane : n1 + (lane - 32);           // place in the reference's dictionary order
    F.lt = 0; F.keep = false; F.touched = 0ull; F.claimed = 0u;
    F.fast = false; F.sp = -1; F.S = 0ull;
    if (n <= k) { F.mode = FR_NOPRUNE; return verify(); }
    CRITP(6);
    // ---- bitonic merge of (weight, home lane): hub entries ascending in lanes [0, n1), +inf up to lane 31 ----
    double sw = lane < 32 ? (in1 ? w1 : __longlong_as_double(0x7ff0000000000000ll)) : B.sw;
    int sid = lane < 32 ? lane : B.sid;
    merge_stage<32>(sw, sid);
    merge_stage<16>(sw, sid);
    merge_stage<8>(sw, sid);
    merge_stage<4>(sw, sid);
    merge_stage<2>(sw, sid);
    merge_stage<1>(sw, sid);
    // ---- runs of equal weights: a run starts where a lane differs from its left neighbour ----
    const long long swb = __double_as_longlong(sw);
    const int llo = __builtin_amdgcn_mov_dpp((int)(unsigned)(swb & 0xffffffffll), 0x138, 0xf, 0xf, true);   // wave_shr:1 (lane 0: 0)
    const int lhi = __builtin_amdgcn_mov_dpp((int)(swb >> 32), 0x138, 0xf, 0xf, true);
    const double left = __longlong_as_double(((long long)lhi << 32) | (unsigned)llo);
    const u64 S = __ballot(left != sw) | 1ull;
    // number of strictly smaller candidates of the candidate at sorted lane p = lane of its run's start
    const u64 below = S & (((u64)2 << lane) - 1ull);            // run starts at or below this lane (never empty: bit 0)
    const int rs = 63 - __builtin_clzll(below);
    const int both = push_i32(rs | (lane << 8), sid);           // ... carried, with the sorted position, to the lane the candidate lives in
    const int lt = both & 0xff;
    F.fast = true; F.sp = both >> 8; F.S = S;
    const int drop = n - k;
    const bool mine = (F.live >> lane) & 1ull;
    F.lt = lt;
    F.keep = mine && lt >= drop;
    const bool full = (S >> drop) & 1ull;                       // the cut falls on a run start: exactly k candidates are kept
    const unsigned claimed = (unsigned)(S >> drop) & ((1u << k) - 1u);
    F.claimed = claimed;
    F.mode = full ? (claimed == (1u << k) - 1u ? FR_RANKS : FR_TIES) : FR_STRADDLE;
    CRITP(7);
    return verify();
}

// The ORDER half: posA = dictionary position of this lane's s1 entry (lanes < n1).  Returns the slot of this
// lane's candidate in the new dictionary (-1: dropped / none); *n_new = its length.
__device__ inline int merge_order(WaveLds &L, int lane, int k, Front &F, int posA, int *n_new, int g_stamp_i = -1)
{
    const bool mine = (F.live >> lane) & 1ull;
    const int pos = lane < 32 ? posA : F.pos_tail;
    const int n = F.n;
    if (F.mode == FR_NOPRUNE) { *n_new = n; return mine ? pos : -1; }
    *n_new = k;
    int slot;
    if (F.mode == FR_RANKS) {
        slot = F.keep ? F.lt - (n - k) : -1;
    } else if (F.mode == FR_NAN) {
        // a NaN weight (only ever from imported state): numba's lt() orders NaNs by the quicksort's
        // dynamics -> the general selection over LDS, on the list in dictionary order
        int *lane_at = reinterpret_cast<int *>(L.key);              // candidate lane at every list position
        if (mine) { lane_at[pos] = lane; L.w[pos] = F.w; }
        wave_sync();
        (void)topk_select_wave(L.w, n, k, L.sel, L.sort, L.sort.r, L.sort.stk);
        const int who = lane < k ? lane_at[L.sel[lane]] : 63;       // candidate lane that takes slot `lane`
        wave_sync();
        const int got = push_i32(lane < k ? lane + 1 : 0, who);
        slot = mine && lane != 63 ? got - 1 : -1;
    } else {
        slot = ties_order(F.lt, F.live, pos, n, k, L.sort);       // :553-559 (second half)
    }
    STAMP2(5);
    return slot;
}

// both halves (s1's row is in dictionary order: position = lane)
__device__ inline int merge_pair_reg(WaveLds &L, int lane, int k, double alpha, double beta, const Row &r1,
                                     const Row &r2, u64 newkey, double newts, Cand &out, int pre = 0,
                                     int g_stamp_i = -1)
{
    Front F;
    merge_front(L, lane, k, alpha, beta, r1, r2, newkey, newts, F, pre, g_stamp_i);
    int n_new;
    out.slot = merge_order(L, lane, k, F, lane, &n_new, g_stamp_i);
    out.key = F.key; out.ts = F.ts; out.w = F.w;
    return n_new;
}

// Write a whole row (all k entries, zeros beyond n) with one tag.
__device__ __forceinline__ void store_row_at(u64 *base, int k, int lane, int n, u64 key, double ts, double w,
                                             double new_norm, unsigned tag);
__device__ __forceinline__ void store_row(const zt_tppr &h, int m, long long x, int lane, int n, u64 key, double ts,
                                          double w, double new_norm, unsigned tag)
{
    store_row_at(h.rows + ((long long)m * h.N + x) * h.rg, h.k, lane, n, key, ts, w, new_norm, tag);
}
__device__ __forceinline__ void store_row_at(u64 *base, int k, int lane, int n, u64 key, double ts, double w,
                                             double new_norm, unsigned tag)
{
    if (lane < k) {
        const bool a = lane < n;
        const u64 kk = a ? key : 0ull;
        const u64 tt = a ? (u64)__double_as_longlong(ts) : 0ull;
        const u64 ww = a ? (u64)__double_as_longlong(w) : 0ull;
        u64 *e = base + HDR + lane;
        st_agent(e, granule(tag, (unsigned)kk));
        st_agent(e + k, granule(tag, (unsigned)(kk >> 32)));
        st_agent(e + 2 * k, granule(tag, (unsigned)tt));
        st_agent(e + 3 * k, granule(tag, (unsigned)(tt >> 32)));
        st_agent(e + 4 * k, granule(tag, (unsigned)ww));
        st_agent(e + 5 * k, granule(tag, (unsigned)(ww >> 32)));
    }
    if (lane < 3) {
        const u64 nn = (u64)__double_as_longlong(new_norm);
        const unsigned pay = lane == 0 ? (unsigned)n : (lane == 1 ? (unsigned)nn : (unsigned)(nn >> 32));
        st_agent(base + lane, granule(tag, pay));
    }
}

// The same from a Cand (merge_pair_reg): every candidate lane writes its own entry into its slot; slots
// [n, k) are zeroed by lanes n..k-1.
__device__ __forceinline__ void store_row_scatter_at(u64 *base, int k, int lane, int n, const Cand &c, double new_norm,
                                                     unsigned tag);
__device__ __forceinline__ void store_row_scatter(const zt_tppr &h, int m, long long x, int lane, int n, const Cand &c,
                                                  double new_norm, unsigned tag)
{
    store_row_scatter_at(h.rows + ((long long)m * h.N + x) * h.rg, h.k, lane, n, c, new_norm, tag);
}
__device__ __forceinline__ void store_row_scatter_at(u64 *base, int k, int lane, int n, const Cand &c, double new_norm,
                                                     unsigned tag)
{
    if (c.slot >= 0) {
        const u64 tt = (u64)__double_as_longlong(c.ts), ww = (u64)__double_as_longlong(c.w);
        u64 *e = base + HDR + c.slot;
        st_agent(e, granule(tag, (unsigned)c.key));
        st_agent(e + k, granule(tag, (unsigned)(c.key >> 32)));
        st_agent(e + 2 * k, granule(tag, (unsigned)tt));
        st_agent(e + 3 * k, granule(tag, (unsigned)(tt >> 32)));
        st_agent(e + 4 * k, granule(tag, (unsigned)ww));
        st_agent(e + 5 * k, granule(tag, (unsigned)(ww >> 32)));
    }
    if (lane >= n && lane < k) {
        u64 *e = base + HDR + lane;
#pragma unroll
        for (int q = 0; q < 6; ++q) st_agent(e + q * k, granule(tag, 0u));
    }
    if (lane < 3) {
        const u64 nn = (u64)__double_as_longlong(new_norm);
        const unsigned pay = lane == 0 ? (unsigned)n : (lane == 1 ? (unsigned)nn : (unsigned)(nn >> 32));
        st_agent(base + lane, granule(tag, pay));
    }
}

// A wait gave up: set the status word and log what was being waited for (the first CTL_LOG reports of
// a launch are kept at ctl[16 + 8 * slot]; zt_tppr_status prints them).  status = ctl + 2.
__device__ __noinline__ void note_timeout(int *status, int kind, int a, int b, int c, int d)
{
    if (lane_id() != (int)__builtin_ctzll(__ballot(1))) return;   // first active lane reports
    int *ctl = status - 2;
    const int slot = atomicAdd(ctl + 13, 1);
    if (slot < CTL_LOG) {
        int *r = ctl + 16 + 8 * slot;
        r[0] = kind; r[1] = a; r[2] = b; r[3] = c; r[4] = d; r[5] = (int)blockIdx.x; r[6] = (int)(threadIdx.x / WAVE);
        r[7] = (int)(wall_clock64() >> 10);
    }
    __threadfence();
    atomicExch(status, ZT_ERR_TIMEOUT);
    latch_failure(*reinterpret_cast<int **>(ctl + 14), ZT_ERR_TIMEOUT);    // ctl[14..15]: address of the handle's latch
}

// One wait of the launch has already timed out: the others stop waiting too (their results are void).
__device__ __forceinline__ bool launch_failed(const int *status) { return ld_agent(status) == ZT_ERR_TIMEOUT; }

// Spin until flag == epoch (bounded).  Returns false on timeout.
__device__ __forceinline__ bool wait_flag(const unsigned *flag, unsigned epoch, int *status, int what)
{
    unsigned spins = 0;
    long long t0 = 0;
    while (ld_agent(flag) != epoch) {
        __builtin_amdgcn_s_sleep(4);
        if ((++spins & 1023u) == 0) {
            const long long now = (long long)wall_clock64();
            if (t0 == 0) t0 = now;
            else if (now - t0 > WAIT_TICKS) { note_timeout(status, 1, what, (int)epoch, (int)ld_agent(flag), 0); return false; }
            if (launch_failed(status)) return false;
        }
    }
    return true;
}

// Load a row that an earlier edge of this launch rewrites: poll until every
// granule carries `expect`.  The mismatching tag tells how many chain hops are
// still ahead, which sets the back-off.
// poll a row until its granules carry the expected tag.  `version`: the row is a slot of a chain's hub-row versions (all
// of a launch's versions carry one tag, so there is no "hops still to go" to size the naps by: short, even naps)
__device__ inline bool load_row_wait_at(const u64 *base, int k, int lane, unsigned expect, Row &r, int *status, int x, int m,
                                        bool version, unsigned *last_seen = nullptr)
{
    unsigned polls = 0;
    long long t0 = 0;
    for (;;) {
        const unsigned seen = load_row_at(base, k, lane, expect, r);
        if (last_seen) *last_seen = seen;
        if (seen == expect) return true;
        if (version) {
            __builtin_amdgcn_s_sleep(6);
        } else {
            // hops still to go on this node's chain (tags of older launches count as ordinal 0)
            const unsigned cur = (seen >> ORD_BITS) == (expect >> ORD_BITS) ? (seen & ((1u << ORD_BITS) - 1)) : 0u;
            const unsigned want = expect & ((1u << ORD_BITS) - 1);
            int ahead = (int)want - (int)cur - 1;               // 0: my predecessor is being written right now
            if (ahead > 0) {
                int naps = ahead > 64 ? 64 : ahead;             // ~1.5 us per hop ahead, capped
                for (int q = 0; q < naps; ++q) __builtin_amdgcn_s_sleep(56);
            } else {
                __builtin_amdgcn_s_sleep(2);
            }
        }
        if ((++polls & 255u) == 0) {
            const long long now = (long long)wall_clock64();
            if (t0 == 0) t0 = now;
            else if (now - t0 > WAIT_TICKS) { note_timeout(status, version ? 4 : 2, x, (int)expect, (int)seen, m); return false; }
            if (launch_failed(status)) return false;
        }
    }
}
__device__ inline bool load_row_wait(const zt_tppr &h, int m, long long x, int lane, unsigned expect, Row &r,
                                     int *status, unsigned *last_seen = nullptr)
{
    return load_row_wait_at(h.rows + ((long long)m * h.N + x) * h.rg, h.k, lane, expect, r, status, (int)x, m, false, last_seen);
}


}  // namespace
