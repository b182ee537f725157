// TWO hops of a hub chain in ONE critical section (round 5).
//
// The reference applies a batch's edges one by one (utils/util.py:495-574): the hub's row after edge t is the input of
// edge t + 1, and the chain's time is (hops) x (critical section).  Two consecutive hops t, t + 1 with partners p1, p2:
//     R1 = prune_k( R s1  (+) Q1 s2  (+) x1 )            (utils/util.py:514-564 for the pair (hub, p1))
//     R2 = prune_k( R1 s1' (+) Q2 s2' (+) x2 )           (... (hub, p2), on the norm hop t leaves)
// When no key is shared between R, Q1's side and Q2's side, R2 is the top k of the 3k + 2 candidates
//     R s1 s1'  |  Q1 s2 s1', x1 s1'  |  Q2 s2', x2
// (each weight rounded exactly as the two hops round it): an entry the first prune drops has k others above it that are
// scaled by the same s1' afterwards, so it cannot come back -- except through a tie with a cut.  At k = 20 that is
// 20 + 21 + 21 = 62 candidates: one per lane, ONE 64-lane bitonic merge network (the hub's row arrives ascending, the
// two partner sides are merged and sorted descending while the wave still waits for the row), and the rank of a
// candidate inside either hop's candidate set is a population count over the sorted positions.  The section publishes R2
// in the chain's usual two-stage form (tppr_hop.hpp: provisional arrangement, runs of equal weights flagged, a straddling
// run as picks + alternates); R1 -- needed by edge t + 1's partner task and by the hub's readers as version t + 1 -- and
// the dictionary orders of both rows (numba's argsort, replayed) are worked out by the same wave OFF the chain, one hop
// after the other, with the code the single hop uses.
//
// Preconditions, checked where they can be (a failed check before anything is written leaves both positions to the
// single hop, run by this wave one after the other):
//   at the claim      k <= 20; both edges have a partner other than the hub, and not the same one;
//   in preparation    the hub's norm is predictable (norm <- norm beta + beta from hop to hop); neither new key is in a
//                     row already; Q1's and Q2's keys are disjoint and free of slot collisions in the wave's hash table;
//   in the section    the row arrived sorted, full, with the predicted norm; no key of R is in Q1 or Q2; hop t's cut does
//                     not fall inside a run of equal weights that reaches R2 (then which members survive hop t would
//                     matter); a picked member of the PREVIOUS hop's straddling run is dropped for certain by hop t.
// How often it applies is counted in the launch's control words (zt_tppr_chain_stats).
#pragma once

#include "tppr_chain.hpp"

namespace {

constexpr int PAIR_K_MAX = 20;       // k + 2 (k + 1) <= 62: lane 63 is the lane nobody reads (tppr_rows.hpp)
constexpr int PA = 20, PB = 41;      // home lanes: the hub's entries 0 .. k-1, partner 1's side PA + j, partner 2's side PB + j
constexpr int HTAB2 = HTAB / 2;      // the wave's hash table in two halves: partner 1's keys in slots [0, HTAB2), partner 2's in [HTAB2, HTAB) (9-bit slots)

// What the paired hop needs of the handle and the launch, by VALUE: the function is a real call (inlined beside the single
// hop and the general task it pushed the kernel past 256 registers), and a reference to the kernel's argument structs
// would make the compiler keep copies of them in scratch memory.
struct PairCtx {
    u64 *rows, *hubver;
    unsigned *cdone;
    int *ctl;
    const double *tsv;
    const long long *eidx;
    long long N;
    double alpha, beta;
    unsigned epoch;
    int k, rg, m;
};
__device__ __forceinline__ u64 *pair_version(const PairCtx &X, int c, int t)
{
    return X.hubver + (((size_t)X.m * MAX_CHAINS + c) * (CH_MAX + 1) + t) * X.rg;
}
__device__ __forceinline__ RowSrc pair_row_src(const PairCtx &X, long long x, int wo, int hv, unsigned tag_base, unsigned vtag)
{
    RowSrc s;
    s.slot = wo;
    if (hv >= 0) { s.base = pair_version(X, hv, wo); s.expect = vtag; s.version = true; s.polled = true; }
    else { s.base = X.rows + ((long long)X.m * X.N + x) * X.rg; s.expect = wo ? (tag_base | (unsigned)wo) : 0u; s.version = false; s.polled = wo != 0; }
    return s;
}

// A row loaded into lanes off .. off + k - 1 (entry j in lane off + j); the header granules by lanes 0 .. 2 as always.
__device__ __forceinline__ void load_row_issue_off(const u64 *base, int k, int lane, int off, RawRow &q)
{
    q.g0 = q.g1 = q.g2 = q.g3 = q.g4 = q.g5 = q.gh = 0;
    if (lane < 3) q.gh = ld_agent(base + lane);
    const int e = lane - off;
    if ((unsigned)e < (unsigned)k) {
        const u64 *p = base + HDR + e;
        q.g0 = ld_agent(p);
        q.g1 = ld_agent(p + k);
        q.g2 = ld_agent(p + 2 * k);
        q.g3 = ld_agent(p + 3 * k);
        q.g4 = ld_agent(p + 4 * k);
        q.g5 = ld_agent(p + 5 * k);
    }
}
__device__ __forceinline__ unsigned row_from_raw_off(const RawRow &q, int k, int lane, int off, unsigned expect, Row &r)
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
    if ((unsigned)(lane - off) < (unsigned)k) {
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
// ... polled until every granule carries `expect` (see load_row_wait_at)
__device__ inline bool load_row_wait_off(const RowSrc &src, int k, int lane, int off, Row &r, int *status, int x, int m)
{
    unsigned polls = 0;
    long long t0 = 0;
    for (;;) {
        RawRow q;
        load_row_issue_off(src.base, k, lane, off, q);
        const unsigned seen = row_from_raw_off(q, k, lane, off, src.expect, r);
        if (seen == src.expect || !src.polled) return true;
        if (src.version) __builtin_amdgcn_s_sleep(6); else __builtin_amdgcn_s_sleep(2);
        if ((++polls & 255u) == 0) {
            const long long now = (long long)wall_clock64();
            if (t0 == 0) t0 = now;
            else if (now - t0 > WAIT_TICKS) { note_timeout(status, src.version ? 4 : 2, x, (int)src.expect, (int)seen, src.aux(m)); return 0.0; }
            if (launch_failed(status)) return 0.0;
        }
    }
}

__device__ __forceinline__ u64 bits_upto(int b) { return ((u64)2 << b) - 1ull; }     // positions 0 .. b (b <= 63)
__device__ __forceinline__ u64 bits_in(int a, int b) { return bits_upto(b) & ~bits_upto(a); }   // positions a + 1 .. b

// Chain positions t (edge i1, partner rec1) and t + 1 (edge i2, rec2) of hub `hub` by ONE wave in ONE critical section.
// Returns the hub's norm after hop t + 1 (> 0), or 0.0 -- nothing of either hop done, the mailbox untouched, the wave's
// hash table clean -- when a precondition fails: the caller runs the two single hops.  next_edge = the chain's edge at
// t + 2 (or -1); hint_norm / hint_tpos: the wave's last hop (ChainHint).
__device__ __attribute__((noinline)) double chain_hop2(PairCtx X, WaveLds *Lp, int lane, int i1, int i2, Mail *mail, long long hub,
                                                      int next_edge, int t, double hint_norm, int hint_tpos, int chain_idx,
                                                      HopRec rec1, HopRec rec2)
{
    WaveLds &L = *Lp;
#ifdef ZT_PAIR_STAT
    const long long ps_t0 = (long long)__builtin_readcyclecounter();
#define PS_ADD(w, a, b) do { if (lane == 0) atomicAdd((unsigned long long *)&g_pstat[(w)], (unsigned long long)((b) - (a))); } while (0)
#else
#define PS_ADD(w, a, b) do { } while (0)
#endif
    const int k = X.k;
    const int m = X.m;
    const double alpha = X.alpha, beta = X.beta;
    const unsigned epoch = X.epoch, tag_base = epoch << ORD_BITS, vtag = tag_base | 1u;
    int *status = X.ctl + 2;
    int wl_fail = 0;
    // ---- both partners' rows, each into its home lanes, on their way while the rest is looked up ----
    const long long p1 = rec1.partner, p2 = rec2.partner;
    const RowSrc src1 = pair_row_src(X, p1, rec1.wo_p, rec1.pchain, tag_base, vtag);
    const RowSrc src2 = pair_row_src(X, p2, rec2.wo_p, rec2.pchain, tag_base, vtag);
    RawRow raw1, raw2;
    load_row_issue_off(src1.base, k, lane, PA, raw1);
    load_row_issue_off(src2.base, k, lane, PB, raw2);
    const double tnow1 = X.tsv[i1], tnow2 = X.tsv[i2];
    const long long e1 = X.eidx[i1], e2 = X.eidx[i2];
    const u64 x1 = ((u64)(unsigned)e1 << 32) | (u64)(unsigned)p1;     // (edge_idx, partner, ts) enters the hub's dictionary
    const u64 x2 = ((u64)(unsigned)e2 << 32) | (u64)(unsigned)p2;
    MailSlot *in_slot = &mail->slot[(t - 1) % MAIL_R], *mid_slot = &mail->slot[t % MAIL_R], *out_slot = &mail->slot[(t + 1) % MAIL_R];
    // ---- the norm the hub's row will arrive with (chain_hop): from this wave's last hop, or the latest kept set in the ring ----
    double pn = 0.0;
    int psteps = -1;
    if (hint_tpos >= 0 && t - hint_tpos <= 24) { pn = hint_norm; psteps = t - hint_tpos - 1; }
    else {
        for (int d = 2; d < MAIL_R && t - d >= 0; ++d) {
            const MailSlot *sl = &mail->slot[(t - d) % MAIL_R];
            if (lds_load_seq(&sl->seq_set) == t - d + 1) {
                double hn; int a0, a1, a2, a3; unsigned a4;
                mail_hdr_read(sl, hn, a0, a4, a1, a2, a3);
                pn = hn; psteps = d - 1;
                break;
            }
        }
    }
    Row rp1, rp2;
    if (row_from_raw_off(raw1, k, lane, PA, src1.expect, rp1) != src1.expect && src1.polled)
        if (!load_row_wait_off(src1, k, lane, PA, rp1, status, (int)p1, m)) wl_fail |= 2;
    // Partner 2's row must NOT be waited for: its last writer may be an edge between the two positions that reads the hub's
    // row as a negative sample -- version t + 1, which this wave produces only after the section (the single hop publishes
    // it before it looks at partner 2).  A few polls; not there: the single hops.
    bool have2 = !src2.polled || row_from_raw_off(raw2, k, lane, PB, src2.expect, rp2) == src2.expect;
    for (int tries = 0; !have2 && tries < 6; ++tries) {
        __builtin_amdgcn_s_sleep(16);
        load_row_issue_off(src2.base, k, lane, PB, raw2);
        have2 = row_from_raw_off(raw2, k, lane, PB, src2.expect, rp2) == src2.expect;
    }
    if (!src2.polled) (void)row_from_raw_off(raw2, k, lane, PB, 0u, rp2);
    // ("the chain has read the partner's old row" -- cdone, which lets the partner task store the new one -- is said only once
    //  the pair is through its last check: the single hops a failed pair falls back to read the rows again)
    if (psteps < 0 || wl_fail || !have2) { chain_stat(X.ctl, lane, ST_PAIR_BAIL_PREP); return 0.0; }
    for (int q = 0; q < psteps; ++q) pn = pn * beta + beta;
    if (pn == 0.0) { chain_stat(X.ctl, lane, ST_PAIR_BAIL_PREP); return 0.0; }
    const double nn = pn * beta + beta, nn2 = nn * beta + beta;                 // the norm after hop t, after hop t + 1 (:570-574)
    const double s1 = pn / nn * beta, s2 = beta / nn * (1.0 - alpha);           // hop t (:520-527)
    const double s1b = nn / nn2 * beta, s2b = beta / nn2 * (1.0 - alpha);       // hop t + 1, on the norm hop t leaves
    const int len1 = __builtin_amdgcn_readfirstlane((rp1.norm != 0.0) ? rp1.len : 0);
    const int len2 = __builtin_amdgcn_readfirstlane((rp2.norm != 0.0) ? rp2.len : 0);
    const int nb1 = len1 + 1, nb2 = len2 + 1, nb12 = nb1 + nb2;
    const bool isq1 = lane >= PA && lane < PA + len1, isx1 = lane == PA + len1;
    const bool isq2 = lane >= PB && lane < PB + len2, isx2 = lane == PB + len2;
    const bool isb = isq1 || isx1 || isq2 || isx2;
    // ---- this lane's candidate (the hub's lanes are filled when the row arrives): key, time, weight after hop t (cw1:
    // ---- what version t + 1 holds) and after hop t + 1 (cw2) ----
    u64 ckey = isq1 ? rp1.key : (isq2 ? rp2.key : (isx1 ? x1 : x2));
    double cts = isq1 ? rp1.ts : (isq2 ? rp2.ts : (isx1 ? tnow1 : tnow2));
    double cw1 = isq1 ? rp1.w * s2 : ((alpha != 0.0) ? s2 * alpha : s2);       // :530-541
    double cw2 = (isq1 || isx1) ? cw1 * s1b : (isq2 ? rp2.w * s2b : ((alpha != 0.0) ? s2b * alpha : s2b));
    {
        // a new key that is in a row already, the same key on both sides, a NaN: the single hops sort it out
        bool bad = isb && (cw2 != cw2 || ((isq1 || isx1) && cw1 != cw1));
        bad = bad || ((isq1 || isq2) && ((ckey == x1 && cts == tnow1) || (ckey == x2 && cts == tnow2)));
        bad = bad || (x1 == x2 && tnow1 == tnow2);
        if (__ballot(bad) != 0ull) { chain_stat(X.ctl, lane, ST_PAIR_BAIL_PREP); return 0.0; }
    }
    // ---- hash table: partner 1's keys in slots [0, HTAB2), partner 2's in [HTAB2, HTAB), value = home lane ----
    int code1 = 0, code2 = 0, slot1h = 0, slot2h = 0;
    if (len1 > 0) {
        code1 = -1;
#pragma unroll
        for (int var = 0; var < 3; ++var) {
            const int hs = key_hash_by(ckey, 2 * var + 1) >> 1;
            if (isq1) L.htab[hs] = lane;
            wave_sync();
            const bool clash = __ballot(isq1 && L.htab[hs] != lane) != 0ull;
            if (clash && isq1) L.htab[hs] = -1;
            wave_sync();
            if (!clash) { code1 = 2 * var + 1; slot1h = hs; break; }
        }
    }
    auto clear1 = [&]() { if (code1 > 0 && isq1) L.htab[slot1h] = -1; };
    if (code1 < 0) { chain_stat(X.ctl, lane, ST_PAIR_BAIL_PREP); return 0.0; }
    if (len2 > 0) {
        // is a key of partner 2 in partner 1's row?  (an occupied slot: compare in full)
        if (code1 > 0) {
            const int c = isq2 ? L.htab[key_hash_by(ckey, code1) >> 1] : -1;
            bool hit = false;
            if (__ballot(c >= 0) != 0ull) {
                const int srcl = c >= 0 ? c : 0;
                const u64 kj = __shfl(ckey, srcl);
                const double tj = __shfl(cts, srcl);
                hit = c >= 0 && kj == ckey && tj == cts;
            }
            if (__ballot(hit) != 0ull) { clear1(); wave_sync(); chain_stat(X.ctl, lane, ST_PAIR_BAIL_PREP); return 0.0; }
        }
        code2 = -1;
#pragma unroll
        for (int var = 0; var < 3; ++var) {
            const int hs = HTAB2 + (key_hash_by(ckey, 2 * var + 1) >> 1);
            if (isq2) L.htab[hs] = lane;
            wave_sync();
            const bool clash = __ballot(isq2 && L.htab[hs] != lane) != 0ull;
            if (clash && isq2) L.htab[hs] = -1;
            wave_sync();
            if (!clash) { code2 = 2 * var + 1; slot2h = hs; break; }
        }
        if (code2 < 0) { clear1(); wave_sync(); chain_stat(X.ctl, lane, ST_PAIR_BAIL_PREP); return 0.0; }
    }
    auto clear_tables = [&]() {
        clear1();
        if (code2 > 0 && isq2) L.htab[slot2h] = -1;
        wave_sync();
    };
    // ---- the two partner sides as ONE descending sequence behind +inf padding (lanes 64 - nb12 .. 63), each weight with
    // ---- the lane its candidate lives in: rb = candidates of the sides that come before mine ----
    double sw;
    int sid;
    {
        const u64 bmask = __ballot(isb);
        int rb = 0;
        u64 todo = bmask;
        while (todo != 0ull) {                                      // (<= 42 rounds of scalar lane reads)
            const int q = __ffsll((long long)todo) - 1;
            todo &= todo - 1ull;
            const double x = readlane_f64(cw2, q);
            rb += (x > cw2 || (x == cw2 && q < lane)) ? 1 : 0;
        }
        const int dst = isb ? 64 - nb12 + rb : 0;                   // (lanes without a candidate push to lane 0: a hub lane, filled later)
        sw = push_f64(cw2, dst);
        sid = push_i32(lane, dst);
        if (lane < 64 - nb12) { sw = __longlong_as_double(0x7ff0000000000000ll); sid = 63; }     // padding (lanes < k: the hub's, below)
    }
#ifdef ZT_PAIR_STAT
    const long long ps_tw = (long long)__builtin_readcyclecounter();
    PS_ADD(0, ps_t0, ps_tw);                                                    // preparation (incl. the partners' rows)
#endif
    // waves whose turn is two or more hops away doze (chain_hop)
    if (t >= 2) {
        const int *far = &mail->slot[(t - 2) % MAIL_R].seq_set;
        unsigned spins = 0;
        while (lds_load_seq(far) != t - 1 && lds_load_seq(&in_slot->seq_set) != t) {
            __builtin_amdgcn_s_sleep(8);
            if ((++spins & 1023u) == 0 && launch_failed(status)) break;
        }
    }
    if (!wait_seq(&in_slot->seq_set, t, status, i1, -3, true)) { clear_tables(); return 0.0; }
    __builtin_amdgcn_s_setprio(3);
#ifdef ZT_PAIR_STAT
    const long long ps_t1 = (long long)__builtin_readcyclecounter();
    PS_ADD(2, ps_tw, ps_t1);                                                    // waiting for the turn
#endif
    // ================= the critical section =================
    double hn;
    int hlen_v, hmunc_v, hnalt_v, hsorted_v;
    unsigned hunc_v;
    mail_hdr_read(in_slot, hn, hlen_v, hunc_v, hmunc_v, hnalt_v, hsorted_v);
    const bool ishub = lane < k;
    double hw = 0.0;
    if (lane < 32) { const u64 kk = in_slot->key[lane]; const double tt = in_slot->ts[lane]; hw = in_slot->w[lane]; if (ishub) { ckey = kk; cts = tt; } }
    const int fs_out = lds_load_seq(&out_slot->seq_free), fs_mid = lds_load_seq(&mid_slot->seq_free);
    const int n1 = __builtin_amdgcn_readfirstlane(hlen_v), munc = __builtin_amdgcn_readfirstlane(hmunc_v);
    const int nalt = __builtin_amdgcn_readfirstlane(hnalt_v);
    const unsigned hunc = (unsigned)__builtin_amdgcn_readfirstlane((int)hunc_v);
    auto bail = [&]() -> double {
        __builtin_amdgcn_s_setprio(1);
        clear_tables();
        chain_stat(X.ctl, lane, ST_PAIR_BAIL_CRIT);
        return 0.0;
    };
    {   // sorted arrangement, full row, the predicted norm
        const long long hb = __double_as_longlong(hn), pb = __double_as_longlong(pn);
        const unsigned h0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)hb), h1 = (unsigned)__builtin_amdgcn_readfirstlane((int)(hb >> 32));
        const unsigned q0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)pb), q1 = (unsigned)__builtin_amdgcn_readfirstlane((int)(pb >> 32));
        if (__builtin_amdgcn_readfirstlane(hsorted_v) == 0 || h0 != q0 || h1 != q1 || n1 != k) return bail();
    }
    if (munc > 0) {
        // members of the previous hop's straddling run that were not picked may turn out to be in the row (chain_hop)
        const u64 alt = in_slot->alt_key[lane & 31];
        bool tt = lane < nalt && (alt == x1 || alt == x2);
        if (code1 > 0) tt = tt || (lane < nalt && L.htab[key_hash_by(alt, code1) >> 1] >= 0);
        if (code2 > 0) tt = tt || (lane < nalt && L.htab[HTAB2 + (key_hash_by(alt, code2) >> 1)] >= 0);
        if (__ballot(tt) != 0ull) return bail();
    }
    // is a key of the hub's row in a partner's row?  Read now, looked at after the network
    const int cand1 = (ishub && code1 > 0) ? L.htab[key_hash_by(ckey, code1) >> 1] : -1;
    const int cand2 = (ishub && code2 > 0) ? L.htab[HTAB2 + (key_hash_by(ckey, code2) >> 1)] : -1;
    if (ishub) { cw1 = hw * s1; cw2 = cw1 * s1b; sw = cw2; sid = lane; }        // t_s1_PPR[key] = value * scale_s1, twice (:524)
#ifdef ZT_PAIR_STAT
    const long long ps_a = (long long)__builtin_readcyclecounter();
    PS_ADD(5, ps_t1, ps_a);                                                     // section: row in registers, checks, probes issued
#endif
    merge_stage<32>(sw, sid);
    merge_stage<16>(sw, sid);
    merge_stage<8>(sw, sid);
    merge_stage<4>(sw, sid);
    merge_stage<2>(sw, sid);
    merge_stage<1>(sw, sid);
#ifdef ZT_PAIR_STAT
    const long long ps_b = (long long)__builtin_readcyclecounter();
    PS_ADD(6, ps_a, ps_b);                                                      // section: the network
#endif
    // ---- lane p now holds the candidate of sorted position p (ascending; the padding behind position n - 1).  What the
    // ---- PUBLICATION needs is worked out here, mostly on the scalar unit; the ranks the two replays need come later ----
    const int n = n1 + nb12, drop1 = nb1, drop2 = nb2;                          // hop t: k + nb1 candidates, hop t + 1: k + nb2
    const long long swb = __double_as_longlong(sw);
    const int llo = __builtin_amdgcn_mov_dpp((int)(unsigned)(swb & 0xffffffffll), 0x138, 0xf, 0xf, true);   // wave_shr:1
    const int lhi = __builtin_amdgcn_mov_dpp((int)(swb >> 32), 0x138, 0xf, 0xf, true);
    const u64 S = __ballot(__longlong_as_double(((long long)lhi << 32) | (unsigned)llo) != sw) | 1ull;     // run starts
    const bool real = lane < n;
    const bool inA = real && sid < PB;                                          // a candidate of hop t (hub row, partner 1's side)
    const u64 MA = __ballot(inA);
    const bool kept1s = inA && mbcnt64(MA) >= drop1;                            // hop t keeps the top k of its candidates (members of
    const u64 K1 = __ballot(kept1s);                                            //  a run that straddles its cut: by position, see below)
    const bool mem2 = kept1s || (real && !inA);                                 // a candidate of hop t + 1
    const u64 M2 = __ballot(mem2);
    const int idx2 = mbcnt64(M2);
    const bool kept2 = mem2 && idx2 >= drop2;
    const u64 K2 = __ballot(kept2);
    __builtin_amdgcn_sched_barrier(0);
    // ---- the tests that were left for after the network ----
    {
        bool bad = ishub && ((ckey == x1 && cts == tnow1) || (ckey == x2 && cts == tnow2) || cw2 != cw2 || cw1 != cw1);
        if (__ballot(cand1 >= 0 || cand2 >= 0) != 0ull) {                       // an occupied slot: compare the keys in full
            const int sa = cand1 >= 0 ? cand1 : 0, sb = cand2 >= 0 ? cand2 : 0;
            const u64 ka = __shfl(ckey, sa), kb = __shfl(ckey, sb);
            const double ta = __shfl(cts, sa), tb = __shfl(cts, sb);
            bad = bad || (cand1 >= 0 && ka == ckey && ta == cts) || (cand2 >= 0 && kb == ckey && tb == cts);
        }
        if (__ballot(bad) != 0ull) return bail();
    }
    const int pk1 = __ffsll((long long)K1) - 1, pd1 = 63 - __builtin_clzll(MA & (((u64)1 << pk1) - 1ull));   // lowest kept / highest dropped of hop t
    const bool clean1 = (S & bits_in(pd1, pk1)) != 0ull;                        // a run starts in (pd1, pk1]: the cut is not inside a run
    const int pk2 = __ffsll((long long)K2) - 1, pd2 = 63 - __builtin_clzll(M2 & (((u64)1 << pk2) - 1ull));
    const bool clean2 = (S & bits_in(pd2, pk2)) != 0ull;
    // hop t's cut inside a run: which members survive it follows from hop t's replay -- it does not matter here iff every
    // member of that run is dropped by hop t + 1 for certain, i.e. the lowest weight R2 keeps is strictly above the run's
    if (!clean1 && !(pk2 > pk1 && (S & bits_in(pk1, pk2)) != 0ull)) return bail();
    // a candidate hop t dropped never lies strictly above R2's cut (k others of hop t's are above it); one that TIES with the
    // cut may sit above position pk2 inside that run: then "index among hop t + 1's candidates" is not "position - const"
    const u64 nmask = bits_upto(n - 1);
    if ((~M2 & nmask & ~(((u64)1 << pk2) - 1ull)) != 0ull) return bail();
    // a picked member of the PREVIOUS hop's straddling run must be dropped by hop t for certain: below the run of hop t's cut
    if (munc > 0) {
        const int rs1 = 63 - __builtin_clzll(S & bits_upto(pk1));
        if ((__ballot(inA && sid < munc) & ~(((u64)1 << rs1) - 1ull)) != 0ull) return bail();
    }
    // ---- R2's provisional arrangement: position p >= pk2 takes slot p - pk2 (every position from pk2 on is a candidate of
    // ---- hop t + 1); a slot is in doubt iff its position shares its run with a neighbour; a run that straddles the cut:
    // ---- picks + alternates (tppr_hop.hpp, MailSlot) ----
    int munc_out = 0, nalt_out = 0, rsG = 0;
    if (!clean2) {
        rsG = 63 - __builtin_clzll(S & bits_upto(pk2));
        const int endG = __ffsll((long long)(S & ~bits_upto(pk2))) - 1;         // (the padding starts a run at position n <= 62)
        munc_out = endG - pk2;
        nalt_out = __popcll(M2 & (((u64)1 << pk2) - 1ull) & ~(((u64)1 << rsG) - 1ull));
    }
    const u64 multi = (~S | ~(S >> 1)) & nmask;                                 // position p shares its run with p - 1 or with p + 1
    const unsigned unc_out = (unsigned)(multi >> pk2) & ((1u << k) - 1u);
    const bool fin = clean2 && unc_out == 0u;                                   // all kept weights distinct: the arrangement IS the order
    const int slot2p = kept2 ? idx2 - drop2 : -1;
    const bool isalt = !clean2 && mem2 && !kept2 && lane >= rsG;
    const int altidx = isalt ? idx2 - (drop2 - nalt_out) : -1;
    // ... told to the lane each candidate lives in
    const int info = push_i32((slot2p + 1) | ((altidx + 1) << 6), real ? sid : 63);
    const int provslot = (info & 63) - 1, my_alt = ((info >> 6) & 63) - 1;
#ifdef ZT_PAIR_STAT
    const long long ps_c = (long long)__builtin_readcyclecounter();
    PS_ADD(7, ps_b, ps_c);                                                      // section: masks, tests, slots sent home
#endif
    if (t + 1 >= MAIL_R && fs_out != t + 1 - MAIL_R + 1) {
        if (!wait_seq(&out_slot->seq_free, t + 1 - MAIL_R + 1, status, i2, -1)) wl_fail |= 64;
    }
#ifdef ZT_PAIR_STAT
    PS_ADD(8, ps_c, (long long)__builtin_readcyclecounter());                   // section: waiting for the ring slot
#endif
    const bool mine = ishub || isb;
    if (mine && provslot >= 0) { out_slot->key[provslot] = ckey; out_slot->ts[provslot] = cts; out_slot->w[provslot] = cw2; }
    if (mine && my_alt >= 0) out_slot->alt_key[my_alt] = ckey;
    if (lane == 0) mail_hdr_write(out_slot, nn2, k, unc_out, munc_out, nalt_out, 1);
    if (fin && mine && provslot >= 0) out_slot->pos[provslot] = provslot;
    asm volatile("" ::: "memory");
    if (lane == 0) __hip_atomic_store(&out_slot->seq_set, t + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (lane == 0 && fin) __hip_atomic_store(&out_slot->seq_ord, t + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __builtin_amdgcn_s_setprio(0);                                              // the rest of both hops is off the chain
#ifdef ZT_PAIR_STAT
    const long long ps_t2 = (long long)__builtin_readcyclecounter();
    PS_ADD(1, ps_t1, ps_t2);                                                    // critical sections
#endif
    // ================= off the chain =================
    st_agent(X.cdone + (long long)m * MAX_CHUNK + i1, epoch);                   // both partners' old rows have been used: their
    st_agent(X.cdone + (long long)m * MAX_CHUNK + i2, epoch);                   // partner tasks may store the new ones
    chain_stat(X.ctl, lane, ST_PAIR_DONE);
    clear_tables();
    // the skipped position's slot: its header carries the norm after hop t (waves predict norms from the ring), and the
    // wave of position t + 2 takes "position t published" as the sign to poll without sleeping.  Only once the reader of
    // the slot's previous content has let go of it.
    const bool mid_free = t < MAIL_R || fs_mid == t - MAIL_R + 1;
    if (mid_free && lane == 0) {
        mail_hdr_write(mid_slot, nn, k, 0u, 0, 0, 1);
        asm volatile("" ::: "memory");
        __hip_atomic_store(&mid_slot->seq_set, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // ---- the ranks the replays work on: strictly smaller candidates of hop t (by the weights hop t ranks: cw1) and of hop
    // ---- t + 1 (cw2: what the network sorted by), for the candidate of every sorted position, sent home ----
    const u64 LT = lanemask_lt();
    const int rs = 63 - __builtin_clzll(S & bits_upto(lane));                   // start of my run
    const u64 LTrs = ((u64)1 << rs) - 1ull;                                     // positions strictly below my run
    int lt1 = __popcll(MA & LTrs);
    const int lt2 = __popcll(M2 & LTrs);
    bool multiA = false;                                                        // my run holds another candidate of hop t
    {
        const u64 lo = MA & LT, hi = MA & ~LT & ~((u64)1 << lane);
        if (lo != 0ull) multiA = (63 - __builtin_clzll(lo)) >= rs;
        if (hi != 0ull) multiA = multiA || (S & bits_in(lane, __ffsll((long long)hi) - 1)) == 0ull;
        multiA = multiA && inA;
    }
    const u64 TA = __ballot(multiA);
    if (TA != 0ull) {
        // The network sorted by cw2 = cw1 s1', which keeps hop t's order of its candidates but may round two different cw1
        // to one cw2: inside a run of equal cw2, hop t's rank is refined by cw1 (equal cw1 stay tied, as they are for hop t)
        const double c1s = shfl_f64(cw1, real ? sid : lane);                    // cw1 of the candidate at this sorted position
        u64 todo = TA & ~LTrs;                                                  // (uniform per run; walk my run's members)
        int add = 0;
        const int my_rs = rs;
        u64 all = TA;
        while (all != 0ull) {
            const int q = __ffsll((long long)all) - 1;
            all &= all - 1ull;
            const double x = readlane_f64(c1s, q);
            const int qrs = __builtin_amdgcn_readlane(rs, q);
            add += (multiA && qrs == my_rs && x < c1s) ? 1 : 0;
        }
        (void)todo;
        lt1 += add;
    }
    const int info2 = push_i32(lt1 | (lt2 << 6) | (kept1s ? 1 << 12 : 0), real ? sid : 63);
    const int my_lt1 = info2 & 63, my_lt2 = (info2 >> 6) & 63;
    // ---- hop t: the dictionary order of R1 (numba's argsort over hop t's candidate LIST: the hub's entries in their
    // ---- dictionary order, partner 1's unmatched entries, the new key -- utils/util.py:553-559) ----
    const u64 live1 = (((u64)1 << k) - 1ull) | ((((u64)1 << nb1) - 1ull) << PA);
    const bool mine1 = (live1 >> lane) & 1ull;
    const int pos1p = ishub ? lane : k + (lane - PA);                           // place in the list as it arrived (provisional for the hub's)
    int slot1;
    {
        // do ties reach R1?  (the cut inside a run, or a run with two of hop t's candidates, one of them kept)
        const bool ties1 = !clean1 || (TA & K1) != 0ull;
        if (!ties1) {
            slot1 = (mine1 && my_lt1 >= drop1) ? my_lt1 - drop1 : -1;           // all kept weights distinct: ranks decide
        } else {
            const int sc = ties_order(my_lt1, live1, pos1p, k + nb1, k, L.sort);      // final slot of every list POSITION
            int *sig = L.sel;
            if (mine1) sig[pos1p] = sc;
            wave_sync();
            int hub_pos = lane;                                                 // identities: where my candidate REALLY stood
            if (hunc != 0u) {
                if (!wait_seq(&in_slot->seq_ord, t, status, i1, -5)) wl_fail |= 32;
                hub_pos = in_slot->pos[lane & 31];
            }
            const int truepos = ishub ? hub_pos : pos1p;
            slot1 = (mine1 && truepos >= 0) ? sig[truepos] : -1;
            wave_sync();
        }
    }
    {
        Cand c1;
        c1.key = ckey; c1.ts = cts; c1.w = cw1; c1.slot = slot1;
        store_row_scatter_at(pair_version(X, chain_idx, t + 1), k, lane, k, c1, nn, vtag);     // edge t + 1's partner task reads it
    }
    // ---- hop t + 1: R2's dictionary order, over [R1 in ITS order | partner 2's unmatched entries | the new key] ----
    const u64 live2 = __ballot(mine1 && slot1 >= 0) | ((((u64)1 << nb2) - 1ull) << PB);
    const bool mine2 = (live2 >> lane) & 1ull;
    const int pos2 = lane < PB ? slot1 : k + (lane - PB);
    int slot2;
    if (fin) {
        slot2 = mine2 ? provslot : -1;
    } else {
        slot2 = ties_order(my_lt2, live2, pos2, k + nb2, k, L.sort);
        if (mine && slot2 >= 0) { out_slot->key2[slot2] = ckey; out_slot->ts2[slot2] = cts; }
        if (mine && provslot >= 0) out_slot->pos[provslot] = slot2;
        asm volatile("" ::: "memory");
        if (lane == 0) __hip_atomic_store(&out_slot->seq_ord, t + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    {
        Cand c2;
        c2.key = ckey; c2.ts = cts; c2.w = cw2; c2.slot = mine ? slot2 : -1;
        if (next_edge < 0)                                                      // the chain's last hop: back to `rows`
            store_row_scatter_at(X.rows + ((long long)m * X.N + hub) * X.rg, k, lane, k, c2, nn2, tag_base | (unsigned)(t + 2));
        store_row_scatter_at(pair_version(X, chain_idx, t + 2), k, lane, k, c2, nn2, vtag);
    }
    // both stages of the incoming slot have been read; the skipped slot has no other reader than this wave
    if (lane == 0) __hip_atomic_store(&in_slot->seq_free, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (t >= MAIL_R && !mid_free) (void)wait_seq(&mid_slot->seq_free, t - MAIL_R + 1, status, i1, -6);
    if (lane == 0) __hip_atomic_store(&mid_slot->seq_free, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    (void)wl_fail;
#ifdef ZT_PAIR_STAT
    PS_ADD(3, ps_t2, (long long)__builtin_readcyclecounter());                  // both off-chain halves
    PS_ADD(4, 0ll, 1ll);
#endif
    return nn2;
}

}  // namespace
