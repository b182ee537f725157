// The hub chain's hop (chain_hop: the hub's update only, lean critical section) and the exported partner half
// (process_chain_partner).  DESIGN.md section 5, P1.
#pragma once

#include "tppr_hop.hpp"

namespace {

// chain statistics (ctl[6 ..], zt_tppr_chain_stats): pairs claimed / completed in one section / left to the single hop in
// preparation / in the section; hops taken singly.  Spine mode counts the sections the spine ran under ST_PAIR_DONE and the
// positions it left to their helpers under ST_PAIR_BAIL_CRIT.
#ifdef ZT_PAIR_STAT
__device__ long long g_pstat[16];      // diagnostic build: clocks in preparation / critical sections / waiting / off-chain halves, pairs
#endif
enum { ST_PAIR_CLAIM = 6, ST_PAIR_DONE = 7, ST_PAIR_BAIL_PREP = 8, ST_PAIR_BAIL_CRIT = 9, ST_SINGLE = 10 };
__device__ __forceinline__ void chain_stat(int *ctl, int lane, int which, int n = 1)
{
    if (lane == 0) __hip_atomic_fetch_add(ctl + which, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One hop of a hub chain, the common case, as a function of its own: the hub's row comes through the mailbox from the
// chain's previous edge, the partner is another node.  The chain applies the HUB's update only (the rest of the edge is
// process_chain_partner's), so this is process_edge's mailbox path with everything else taken out -- no row selection by
// role, no third row, no emission: what is left between the arrival of the row and the publication of the new kept set
// is the chain's critical path, and every scalar branch and register move on it is paid 200 times per batch.
// Returns false when the hop is not of this kind (first hop, another writer in between, self-loop): process_edge takes it.
// What a hop needs to know about its edge besides the rows, gathered ONCE per launch by the whole chain workgroup into
// LDS (k_stream): from memory these are three levels of dependent loads (edge -> endpoints -> writer ordinals / reader
// flags) at the start of every hop's preparation.
struct HopRec {
    int partner;       // the other endpoint (-1: self-loop)
    int wo_p;          // ordinal of the last earlier writer of the partner (the tag to expect; the hub's own is the position)
    int pchain;        // the partner is a hub itself and this edge a position of ITS chain too: that chain's index -- its
                       // row is version wo_p there and that chain applies its update --, else -1
};

#ifdef ZT_CHAIN_VARIANTS
// spine / duo mode (one wave runs every critical section; the weights' recurrence on a wave of its own): measured slower,
// kept out of the product -- tools/exp/variants/tppr_spine.hpp, variant builds only
#include "tppr_spine.hpp"
#endif

__device__ __attribute__((always_inline)) inline bool chain_hop(const zt_tppr &h, const StreamArgs &A, WaveLds &L, int lane, int i, int mo, Mail *mail,
                                 long long hub, int prev_edge, int next_edge, int tpos, ChainHint *hint, int chain_idx,
                                 const HopRec &rec, const bool spine = false)
{
    if (prev_edge < 0 || h.k > REG_K_MAX) return false;
    const int k = h.k;
    const int m = A.m_lo + mo;
    if (rec.partner < 0) return false;
    const long long pnode = rec.partner;
    const int wo_h = tpos, wo_p = rec.wo_p;                                  // (chain position = the hub's writer ordinal)
    const double alpha = h.alpha[m], beta = h.beta[m];
    const unsigned epoch = A.epoch, tag_base = epoch << ORD_BITS, vtag = tag_base | 1u;
#ifdef ZT_CRIT
    long long crit_t[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    int wl_fail = 0;
    CRIT(4);
    HSTAMP(0);
    // ---- the partner's row: from `rows` (poll where a writer of this launch precedes us), or -- the partner is a hub and
    // ---- its chain holds this edge too -- the version of its row at that chain's position ----
    Row rp;
    const RowSrc psrc = row_src(h, m, pnode, wo_p, rec.pchain, tag_base, vtag);
    RawRow praw;
    load_row_issue(psrc.base, k, lane, praw);                               // on its way while the rest is looked up
    const double tnow = A.tsv[i];
    const long long e = A.eidx[i];
    const bool hub_to_memory = next_edge < 0;                               // see process_edge: the chain's last hop only
    MailSlot *in_slot = &mail->slot[(tpos - 1) % MAIL_R], *out_slot = &mail->slot[tpos % MAIL_R];
    PreScale pre_scale;
    pre_scale.valid = false;
#ifndef ZT_CHAIN_VARIANTS
    // the norm the hub's row will arrive with and the scale factors that follow from it: worked out for every position when
    // the workgroup started (zt_tppr::hubscale, k_stream) -- four doubles, on their way beside the partner's row; the lean
    // section still holds the norm that ARRIVES against this one, bit for bit
    {
        const double *sc = hub_scale(h, m, chain_idx, tpos);
        const double sn = sc[0], sn1 = sc[1], s1 = sc[2], s2 = sc[3];
        pre_scale.norm = sn; pre_scale.norm_next = sn1; pre_scale.scale_s1 = s1; pre_scale.scale_s2 = s2;
        pre_scale.valid = sn != 0.0;
    }
#else
    // (the two float64 divisions of the scale factors: while the partner's row is on its way)
    // the norm the hub's row will arrive with: norm <- norm * beta + beta from hop to hop, starting from this wave's own
    // last hop or, if that is long ago (or never was), from the latest kept set in the ring
    double pn = 0.0;
    int psteps = -1;
    if (hint->tpos >= 0 && tpos - hint->tpos <= 24) { pn = hint->norm_out; psteps = tpos - hint->tpos - 1; }
    else {
        for (int d = 2; d < MAIL_R && tpos - d >= 0; ++d) {       // (the slot of position tpos - d is not rewritten before my hop)
            const MailSlot *sl = &mail->slot[(tpos - d) % MAIL_R];
            if (lds_load_seq(&sl->seq_set) == tpos - d + 1) {
                double hn; int a0, a1, a2, a3; unsigned a4;
                mail_hdr_read(sl, hn, a0, a4, a1, a2, a3);
                pn = hn; psteps = d - 1;
                break;
            }
        }
    }
    auto set_scale = [&]() {
        for (int q = 0; q < psteps; ++q) pn = pn * beta + beta;
        if (pn != 0.0) {
            const double nn = pn * beta + beta;
            pre_scale.norm = pn;
            pre_scale.norm_next = nn;
            pre_scale.scale_s1 = pn / nn * beta;
            pre_scale.scale_s2 = beta / nn * (1.0 - alpha);
            pre_scale.valid = true;
        }
    };
    if (psteps >= 0) set_scale();
#endif
    if (row_from_raw(praw, k, lane, psrc.expect, rp) != psrc.expect && psrc.polled)
        if (!load_row_wait_at(psrc.base, k, lane, psrc.expect, rp, h.ctl + 2, (int)pnode, psrc.aux(m), psrc.version)) wl_fail |= 2;
    // ---- while the hub's row is on its way: everything that depends on the partner only ----
    int pre_hash = 0;                           // 1 / 3: partner entered into this wave's hash table (slot function 1 / 2),
    int h2slot = 0;                             // 2: its keys collide under both
    const int lenp = (rp.norm != 0.0) ? rp.len : 0;
    if (lenp > 0) {
        pre_hash = 2;
#pragma unroll
        for (int var = 0; var < 3; ++var) {
            const int hs = key_hash_by(rp.key, 2 * var + 1);
            if (lane < lenp) L.htab[hs] = lane;
            wave_sync();
            const int back = lane < lenp ? L.htab[hs] : lane;
            const bool clash = __ballot(lane < lenp && back != lane) != 0ull;
            if (clash && lane < lenp) L.htab[hs] = -1;           // (several lanes may clear one slot)
            wave_sync();
            if (!clash) { pre_hash = 2 * var + 1; h2slot = hs; break; }
        }
    }
    const u64 nkey = ((u64)(unsigned)e << 32) | (u64)(unsigned)pnode;      // (edge_idx, partner, ts) enters the hub's dictionary
    // what the lean section needs as SCALARS, worked out here, off the chain: the multipliers of the slot function the
    // table was filled with, and the bits of the norm the row must arrive with
    unsigned hm0, hm1;
    key_hash_muls(pre_hash, hm0, hm1);
    hm0 = (unsigned)__builtin_amdgcn_readfirstlane((int)hm0); hm1 = (unsigned)__builtin_amdgcn_readfirstlane((int)hm1);
    const long long pnb = __double_as_longlong(pre_scale.norm);
    const unsigned pn0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)pnb), pn1 = (unsigned)__builtin_amdgcn_readfirstlane((int)(pnb >> 32));
    PreB pre_b;
    pre_b.ok = false;
    if (pre_hash != 2) prepare_b(lane, k, alpha, rp, nkey, tnow, pre_scale, pre_b, h2slot);
    // the partner's row has arrived (its tags were looked at): the partner task may store the partner's new row.  (No
    // s_waitcnt vmcnt(0) here: it would also wait for the write-through stores of this wave's previous hop.)
    st_agent(h.cdone + (long long)m * MAX_CHUNK + i, epoch);
    CRIT(5);
    HSTAMP(1);
#ifdef ZT_CRIT
    crit_t[9] = !pre_scale.valid ? 7 : (pre_hash == 2 ? 8 : (!pre_b.ok ? 9 : 0));     // why the partner's side is not prepared
    crit_t[13] = (long long)ld_agent(h.ctl + 1) * 100000 + i;    // head of the general queue (task index) when this hop was ready, and its edge
#endif
    // ---- spine mode: the prepared side goes to the spine wave, which runs the critical section with the hub's row in its
    // ---- registers (chain_spine) and says so -- or leaves the hop to this wave, the old way ----
    PrepHdr *P = &mail->prep[tpos % PREP_R];
#ifdef ZT_CHAIN_VARIANTS
    bool by_spine = false;
    if (spine) {
        if (!pre_scale.valid && pre_hash != 2) {
            // no norm to start from (this wave's first hop of the launch): the latest kept set within the ring's reach --
            // positions before mine are published without me, the spine cannot pass mine
            unsigned spins = 0;
            bool found = false;
            while (!found) {
                for (int d = 1; d < MAIL_R && tpos - d >= 0; ++d) {
                    const MailSlot *sl = &mail->slot[(tpos - d) % MAIL_R];
                    if (lds_load_seq(&sl->seq_set) == tpos - d + 1) {
                        double hn; int a0, a1, a2, a3; unsigned a4;
                        mail_hdr_read(sl, hn, a0, a4, a1, a2, a3);
                        pn = hn; psteps = d - 1; found = true;
                        break;
                    }
                }
                if (found) break;
                __builtin_amdgcn_s_sleep(4);
                if ((++spins & 1023u) == 0 && launch_failed(h.ctl + 2)) break;
            }
            if (found) {
                set_scale();
                if (pre_scale.valid) prepare_b(lane, k, alpha, rp, nkey, tnow, pre_scale, pre_b, h2slot);
            }
        }
        spine_post(P, L, lane, tpos, pre_b, pre_scale, nkey, tnow, lenp, pre_hash);
        by_spine = spine_wait_res(&P->res, tpos, h.ctl + 2, i) == tpos + 1;

    }
#else
    constexpr bool by_spine = false;                 // (spine / duo mode: variant builds only, tools/exp/variants/tppr_spine.hpp)
    (void)spine;
#endif
    // waves whose turn is two or more hops away doze (see process_edge)
    if (!by_spine && tpos >= 2) {
        const int *far = &mail->slot[(tpos - 2) % MAIL_R].seq_set;
        unsigned spins = 0;
        while (lds_load_seq(far) != tpos - 1 && lds_load_seq(&in_slot->seq_set) != tpos) {
            __builtin_amdgcn_s_sleep(8);
            if ((++spins & 1023u) == 0 && launch_failed(h.ctl + 2)) break;
        }
    }
    // (a wave's first hop of a launch has no norm to start from until somebody has published: the kept set two positions
    //  back is out now -- its successor is in its critical section --, which leaves time to prepare the partner's side)
#ifdef ZT_CHAIN_VARIANTS
    if (!spine && !pre_scale.valid && tpos >= 2 && pre_hash != 2) {
        const MailSlot *sl = &mail->slot[(tpos - 2) % MAIL_R];
        if (lds_load_seq(&sl->seq_set) == tpos - 1) {
            double hn; int a0, a1, a2, a3; unsigned a4;
            mail_hdr_read(sl, hn, a0, a4, a1, a2, a3);
            pn = hn; psteps = 1;
            set_scale();
            if (pre_scale.valid) prepare_b(lane, k, alpha, rp, nkey, tnow, pre_scale, pre_b, h2slot);
        }
    }
#endif
    if (!by_spine) {
        if (!wait_seq(&in_slot->seq_set, tpos, h.ctl + 2, i, prev_edge, true)) wl_fail |= 16;
        __builtin_amdgcn_s_setprio(3);
    }
    CRIT(0);
    HSTAMP(2);
    Row rh;
    unsigned hub_unc = 0u;
    int hub_munc = 0, hub_nalt = 0, hub_sorted = 0;
    u64 hub_alt = 0ull;
    int free_seen = 0;
    bool hub_ordered = false, hub_final = false;
    int hub_pos = lane;
    Front F;
    Cand c;
    double new_norm = 0.0;
    unsigned unc_in = 0u, unc_out = 0u;
    int munc_out = 0, nalt_out = 0, n_new = 0, provslot = -1, trueslot = -1, pos_prov = lane;
    bool set_out = false, final_out = false, mine = false;
    auto hub_order = [&]() {
        if (hub_ordered) return;
        if (!wait_seq(&in_slot->seq_ord, tpos, h.ctl + 2, i, -prev_edge - 2)) wl_fail |= 32;
        hub_pos = in_slot->pos[lane & 31];
        hub_ordered = true;
    };
    auto hub_to_dict = [&]() {
        hub_order();
        if (hub_final) return;
        rh.key = in_slot->key2[lane & 31]; rh.ts = in_slot->ts2[lane & 31];
        hub_pos = lane;
        hub_final = true;
        hub_unc = 0u; hub_munc = 0; hub_nalt = 0;
    };
    auto ring_free = [&]() {
        if (tpos >= MAIL_R) {
            if (free_seen == tpos - MAIL_R + 1) asm volatile("" ::: "memory");
            else if (!wait_seq(&out_slot->seq_free, tpos - MAIL_R + 1, h.ctl + 2, i, -1)) wl_fail |= 64;
        }
    };
    auto publish_set = [&](int sidx, int n, double new_norm, unsigned unc, int munc, int n_alt, int sorted) {
        if (sidx >= 0) { out_slot->key[sidx] = c.key; out_slot->ts[sidx] = c.ts; out_slot->w[sidx] = c.w; }
        if (lane == 0) mail_hdr_write(out_slot, new_norm, n, unc, munc, n_alt, sorted);
    };
    auto publish_seq = [&](bool set, bool ord) {           // LDS only, in program order (see process_edge)
        asm volatile("" ::: "memory");
        if (lane == 0 && set) __hip_atomic_store(&out_slot->seq_set, tpos + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (lane == 0 && ord) __hip_atomic_store(&out_slot->seq_ord, tpos + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    // ================= the lean critical section: the common case, written for instruction count =================
    // The row is the sorted arrangement of a pruned row, its norm the predicted one, the partner's side prepared
    // (prepare_b) and disjoint from it: ranks from the merge network, the kept set written to the mailbox from the
    // lanes the candidates live in.  Every test that fails BEFORE anything is written leaves the hop to the general
    // code below, which starts from the mailbox again.
#ifdef ZT_CRIT
#define LEANC(c) do { crit_t[10] = (c); if (lane == 0 && mo == 0) atomicAdd((unsigned long long *)&g_crit[8199 * 16 + (c)], 1ull); } while (0)
#else
#define LEANC(c) do { } while (0)
#endif
    bool lean_done = false;
#ifndef ZT_NO_LEAN
    if (pre_b.ok && (by_spine || !spine)) {          // (a hop the spine left to this wave failed one of the section's tests already)
        lean_done = [&]() -> bool {
            // the header: one 16-byte LDS word, its four words straight to scalar registers (the fields come apart on the scalar unit)
            const mail_v4u hdr = *reinterpret_cast<const mail_v4u *>(&in_slot->norm);
            const bool low = __builtin_amdgcn_inverse_ballot_w64(0xffffffffull);        // lanes 0..31: the hub's entries
            u64 ckey = pre_b.cb_key;
            double cts = pre_b.cb_ts, cw = pre_b.cb_w, hw = 0.0;
            if (low) { ckey = in_slot->key[lane]; cts = in_slot->ts[lane]; hw = in_slot->w[lane]; }
            const int fs = by_spine ? 0 : lds_load_seq(&out_slot->seq_free);
            const unsigned hz = (unsigned)__builtin_amdgcn_readfirstlane((int)hdr.z);
            const unsigned hunc = (unsigned)__builtin_amdgcn_readfirstlane((int)hdr.w);
            const int n1 = (int)(hz & 0xffu), munc = (int)((hz >> 8) & 0xffu), nalt = (int)((hz >> 16) & 0xffu);
            if (!by_spine) {   // sorted arrangement, predicted norm (bit patterns on the scalar unit: both are finite and positive)
                const unsigned h0 = (unsigned)__builtin_amdgcn_readfirstlane((int)hdr.x), h1 = (unsigned)__builtin_amdgcn_readfirstlane((int)hdr.y);
                if ((hz >> 24) == 0u || h0 != pn0 || h1 != pn1 || (h0 | h1) == 0u) { LEANC(1); return false; }
            }
            const int nb = pre_b.nb, n = n1 + nb, drop = n - k;
            if (!by_spine && (n1 <= 0 || drop <= 0)) { LEANC(2); return false; }
            const bool table = lenp > 0;
            u64 S;
            int both;
            if (by_spine) {
                // the spine ran the section on exactly these inputs (its tests passed): the candidates as they stood, and its
                // ranks -- the rest of this hop is the off-chain half
                if (low) cw = hw * pre_scale.scale_s1;
                const u64 sv = P->S;
                S = ((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(sv >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)sv);
                const int sp_ = L.sort.r[lane];                    // (sorted position; the rank is the start of its run)
                both = (63 - __builtin_clzll(S & (((u64)2 << sp_) - 1ull))) | (sp_ << 8);
            } else {
            if (munc > 0) {
                // members of a straddling run that were not picked may turn out to be in the row (see below)
                const u64 alt = in_slot->alt_key[lane & 31];
                bool t = lane < nalt && alt == nkey;
                if (table) t = t || (lane < nalt && L.htab[key_hash_m(alt, hm0, hm1)] >= 0);
                if (__ballot(t) != 0ull) { LEANC(3); return false; }
            }
            // is a key of the hub's row in the partner's row?  Read now, looked at after the network
            const bool in1 = lane < n1;
            const int cand = (table && in1) ? L.htab[key_hash_m(ckey, hm0, hm1)] : -1;
            const double inf = __longlong_as_double(0x7ff0000000000000ll);
            double sw = pre_b.sw;
            int sid = pre_b.sid;
            if (low) { cw = hw * pre_scale.scale_s1; sw = in1 ? cw : inf; sid = lane; }
            merge_stage<32>(sw, sid);
            merge_stage<16>(sw, sid);
            merge_stage<8>(sw, sid);
            merge_stage<4>(sw, sid);
            merge_stage<2>(sw, sid);
            merge_stage<1>(sw, sid);
            const long long swb = __double_as_longlong(sw);
            const int llo = __builtin_amdgcn_mov_dpp((int)(unsigned)(swb & 0xffffffffll), 0x138, 0xf, 0xf, true);   // wave_shr:1
            const int lhi = __builtin_amdgcn_mov_dpp((int)(swb >> 32), 0x138, 0xf, 0xf, true);
            S = __ballot(__longlong_as_double(((long long)lhi << 32) | (unsigned)llo) != sw) | 1ull;     // run starts
            const u64 below = S & (((u64)2 << lane) - 1ull);
            const int rs = 63 - __builtin_clzll(below);
            both = push_i32(rs | (lane << 8), sid);   // (smaller candidates, sorted position) to the candidate's lane
            __builtin_amdgcn_sched_barrier(0);
            // ---- the tests that were left for after the network ----
            bool bad = in1 && ((ckey == nkey && cts == tnow) || cw != cw);
            if (__ballot(cand >= 0) != 0ull) {                   // an occupied slot: compare the keys in full
                const int src = cand >= 0 ? cand : 0;
                const u64 kj = __shfl(rp.key, src);
                const double tj = __shfl(rp.ts, src);
                bad = bad || (in1 && cand >= 0 && kj == ckey && tj == cts);
            }
            if (__ballot(bad) != 0ull) { LEANC(4); return false; }
            }   // (!by_spine)
            CRIT(1);
            const int lt = both & 0xff, sp = both >> 8;
            const bool full = (S >> drop) & 1ull;               // the cut falls on a run start: exactly k candidates are kept
            const unsigned kmask = (1u << k) - 1u;
            const unsigned claimed = (unsigned)(S >> drop) & kmask;
            const int mode = full ? (claimed == kmask ? FR_RANKS : FR_TIES) : FR_STRADDLE;
            const u64 lowdrop = ((u64)2 << drop) - 1ull;        // positions 0 .. drop
            const int rsG = 63 - __builtin_clzll(S & lowdrop);  // start of the run that holds position `drop`
            if (!by_spine && munc > 0) {
                // a picked member of the previous hop's straddling run that is kept here (or ties with the cut) needs the
                // previous hop's replay first: the general code waits for it
                const int thr = full ? drop : rsG;
                if (__ballot(lane < munc && lt >= thr) != 0ull) { LEANC(5); return false; }
            }
            // ---- provisional slots: the candidate at sorted position p >= drop takes slot p - drop ----
            const u64 nmask = ((u64)2 << (n - 1)) - 1ull;       // positions 0 .. n-1 (n <= 63)
            const u64 multi = (~S | ~(S >> 1)) & nmask;         // position p shares its run with p-1 or with p+1
            const unsigned uo = (unsigned)(multi >> drop) & kmask;
            int mo_ = 0, na_ = 0;
            if (!full) {
                const u64 above = S & ~lowdrop;                 // the next run starts here (the padding's at n, at the latest)
                mo_ = __ffsll((long long)above) - 1 - drop;
                na_ = drop - rsG;
            }
            const int ps = sp - drop;
            const bool kept = (unsigned)ps < (unsigned)k;       // (padding lanes sort behind position n-1)
            const double nn = pre_scale.norm_next;
            const bool fin = mode == FR_RANKS;                   // (then uo == 0: all kept weights distinct)
            if (!by_spine) {
            if (tpos >= MAIL_R && fs != tpos - MAIL_R + 1) {
                if (!wait_seq(&out_slot->seq_free, tpos - MAIL_R + 1, h.ctl + 2, i, -1)) wl_fail |= 64;
            }
            if (kept) { out_slot->key[ps] = ckey; out_slot->ts[ps] = cts; out_slot->w[ps] = cw; }
            if (!full && sp >= rsG && sp < drop) out_slot->alt_key[sp - rsG] = ckey;
            if (lane == 0) mail_hdr_write(out_slot, nn, k, uo, mo_, na_, 1);
            if (fin && kept) out_slot->pos[ps] = ps;
            CRIT(2);
            asm volatile("" ::: "memory");
            if (lane == 0) __hip_atomic_store(&out_slot->seq_set, tpos + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (lane == 0 && fin) __hip_atomic_store(&out_slot->seq_ord, tpos + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            CRIT(3);
            HSTAMP(4);
            __builtin_amdgcn_s_setprio(0);                       // the rest of this hop is off the chain
            }
            // ---- what the tail needs ----
            if (table && lane < lenp) L.htab[pre_b.h2] = -1;     // the table is clean again
            hub_unc = hunc; hub_munc = munc; hub_nalt = nalt;
            hub_ordered = hunc == 0u && munc == 0; hub_final = hub_ordered;
            new_norm = nn;
            F.key = ckey; F.ts = cts; F.w = cw;
            F.live = ((1ull << n1) - 1ull) | (((1ull << nb) - 1ull) << 32);
            F.n = n; F.n1 = n1;
            F.pos_tail = lane < 32 ? lane : n1 + (lane - 32);
            mine = (F.live >> lane) & 1ull;
            F.lt = lt; F.keep = mine && lt >= drop; F.touched = 0ull; F.claimed = claimed;
            F.fast = true; F.sp = sp; F.S = S; F.mode = mode;
            c.key = ckey; c.ts = cts; c.w = cw;
            unc_in = hunc; unc_out = uo; munc_out = mo_; nalt_out = na_;
            pos_prov = F.pos_tail;
            n_new = k;
            provslot = (mine && kept) ? ps : -1;
            set_out = true; final_out = fin;
            if (fin) trueslot = provslot;
            return true;
        }();
    }
#endif
    if (!lean_done) {
    // ---- the hub's row: one batch of LDS reads ----
    mail_hdr_read(in_slot, rh.norm, rh.len, hub_unc, hub_munc, hub_nalt, hub_sorted);
    rh.key = in_slot->key[lane & 31]; rh.ts = in_slot->ts[lane & 31]; rh.w = in_slot->w[lane & 31];
    hub_alt = in_slot->alt_key[lane & 31];
    free_seen = lds_load_seq(&out_slot->seq_free);
    hub_ordered = hub_unc == 0u && hub_munc == 0; hub_final = hub_ordered;
    CRIT(8);
    new_norm = rh.norm * beta + beta;
    bool settled = hub_final;                                    // the row is known to be the dictionary
    if (!settled && hub_munc > 0) {
        // members of a straddling run that were not picked may turn out to be in the row (process_edge)
        bool t = lane < hub_nalt && hub_alt == nkey;
        if (pre_hash == 1 || pre_hash == 3 || pre_hash == 5) t = t || (lane < hub_nalt && L.htab[key_hash_by(hub_alt, pre_hash)] >= 0);
        if (__ballot(t) != 0ull || pre_hash == 2) { hub_to_dict(); settled = true; }
    }
    CRIT(10);
#ifdef ZT_CRIT
    crit_t[11] = hub_sorted ? 0 : 5;
#endif
    if (!(hub_sorted && pre_hash != 5 &&
          merge_front_fast(L, lane, k, rh, rp, pre_b, pre_scale, nkey, tnow, pre_hash == 1 || pre_hash == 3 || lenp == 0,
                           pre_hash == 3, F CRIT_PASS))) {
        if (pre_hash == 3 || pre_hash == 5) {                                     // merge_front probes with the first slot function: start it clean
            if (lane < lenp) L.htab[h2slot] = -1;
            wave_sync();
            pre_hash = 0;
        }
        merge_front(L, lane, k, alpha, beta, rh, rp, nkey, tnow, F, pre_hash, -1, &pre_scale CRIT_PASS);
    }
    CRIT(1);
    if (!settled) {
        bool redo = (F.touched & (u64)hub_unc) != 0ull;
        if (!redo && hub_munc > 0) {
            const bool picked = lane < hub_munc;
            if (F.mode == FR_RANKS || F.mode == FR_TIES) redo = __ballot(picked && F.keep) != 0ull;
            else if (F.mode == FR_STRADDLE) {
                const int top_below = wave_max0(((F.live >> lane) & 1ull) && F.lt < F.n - k ? F.lt + 1 : 0) - 1;   // rank of the straddling run
                redo = __ballot(picked && F.lt >= top_below) != 0ull;
            } else redo = true;
        }
        if (redo) {
            hub_to_dict();
            settled = true;
            merge_front(L, lane, k, alpha, beta, rh, rp, nkey, tnow, F, 0, -1);
        }
    }
    CRIT(12);
    unc_in = settled ? 0u : hub_unc;                             // provisional slots of the row as I used it
    c.key = F.key; c.ts = F.ts; c.w = F.w;
    mine = (F.live >> lane) & 1ull;
    pos_prov = lane < 32 ? lane : F.pos_tail;                    // my candidate's place in the list as it arrived
    const int drop = F.n - k;
    n_new = F.n <= k ? F.n : k;
    ring_free();
    CRIT(13);
    if (F.mode == FR_NOPRUNE) {
        provslot = mine ? pos_prov : -1;                         // s1's entries keep their slots, and their doubts
        unc_out = unc_in;
        set_out = true;
    } else if (F.mode == FR_RANKS) {
        provslot = F.keep ? F.lt - drop : -1;                    // all kept weights distinct: nothing provisional
        set_out = true;
    } else if (F.fast) {
        // Ties, from the sorted positions the merge network left: the candidate at position p >= drop takes provisional
        // slot p - drop (ascending by weight, members of a run of equal weights in whatever order the network put them:
        // "arbitrary" is all stage 1 promises); a slot is in doubt iff its run has another member.  A run that straddles
        // the cut has its members at positions >= drop in slots [0, j) -- the pick -- and the others are the alternates.
        // (wave-uniform by construction: say so, or the 64-bit mask arithmetic below runs on the vector unit)
        const u64 S = ((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(F.S >> 32)) << 32) |
                      (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)F.S);
        const int n_u = __builtin_amdgcn_readfirstlane(F.n);
        const u64 nmask = ((u64)2 << (n_u - 1)) - 1ull;                         // positions 0 .. n-1 (n <= 63)
        const u64 multi = (~S | ~(S >> 1)) & nmask;                             // position p shares its run with p-1 or with p+1
                                                                                // (position n, the padding, always starts a run)
        provslot = (mine && F.sp >= drop) ? F.sp - drop : -1;
        unc_out = (unsigned)(multi >> drop) & ((1u << k) - 1u);
        if (F.mode == FR_STRADDLE) {
            const int rsG = 63 - __builtin_clzll(S & (((u64)2 << drop) - 1ull)); // start of the run that holds position `drop`
            const u64 above = S & ~(((u64)2 << drop) - 1ull);                     // the next run starts here (the padding's at n, at the latest)
            const int endG = __ffsll((long long)above) - 1;
            munc_out = endG - drop;
            nalt_out = drop - rsG;
            if (mine && F.sp >= rsG && F.sp < drop) out_slot->alt_key[F.sp - rsG] = c.key;
        }
        set_out = true;
    } else if (F.mode == FR_TIES || F.mode == FR_STRADDLE) {
        // (process_edge: ascending by weight; equal weights take the slots of their run in lane order; a run that
        //  straddles the cut sends its first j members to slots [0, j) and the others along as alternates)
        int ltG = -1, j = 0;
        u64 Gm = 0ull;
        if (F.mode == FR_STRADDLE) {
            ltG = wave_max0(mine && F.lt < drop ? F.lt + 1 : 0) - 1;
            Gm = __ballot(mine && F.lt == ltG);
            j = ltG + __popcll(Gm) - drop;
        }
        const bool certain = mine && F.lt >= drop;
        const int r0 = F.lt - drop;
        provslot = certain ? r0 : -1;
        unsigned ub = 0u;
        u64 todo = __ballot(certain && r0 + 1 < k && ((F.claimed >> ((r0 + 1) & 31)) & 1u) == 0u);
        while (todo != 0ull) {
            const int l = __ffsll((long long)todo) - 1;
            const int rv = __builtin_amdgcn_readlane(r0, l);
            const u64 grp = __ballot(certain && r0 == rv);
            if ((grp >> lane) & 1ull) { provslot = rv + __popcll(grp & lanemask_lt()); ub = 1u << provslot; }
            todo &= ~grp;
        }
        if (F.mode == FR_STRADDLE) {
            const int gi = __popcll(Gm & lanemask_lt());
            const bool member = (Gm >> lane) & 1ull;
            if (member && gi < j) { provslot = gi; ub = 1u << gi; }
            if (member && gi >= j) out_slot->alt_key[gi - j] = c.key;
            munc_out = j;
            nalt_out = __popcll(Gm) - j;
        }
        unc_out = wave_or(ub);
        set_out = true;
    }
    CRIT(14);
    if (set_out) {
        publish_set(provslot, n_new, new_norm, unc_out, munc_out, nalt_out, F.mode != FR_NOPRUNE ? 1 : 0);
        final_out = unc_out == 0u && munc_out == 0 && F.mode != FR_TIES && F.mode != FR_STRADDLE;
        if (final_out) {
            if (provslot >= 0) out_slot->pos[provslot] = provslot;
            trueslot = provslot;
        }
        CRIT(2);
        publish_seq(true, final_out);                            // the successor can start
        CRIT(3);
        HSTAMP(4);
        __builtin_amdgcn_s_setprio(0);                           // the rest of this hop is off the chain
    }
    }   // (!lean_done)
    hint->norm_out = new_norm; hint->tpos = tpos;
    if (!final_out) {
        // ---- my own replay: final slot of every list POSITION (identity-free, see Mail) ----
        const int slot_c = merge_order(L, lane, k, F, lane, &n_new, -1);
        CRIT(8);
        int *sig = L.sel;                                        // final slot by list position
        if (mine) sig[pos_prov] = slot_c;
        wave_sync();
        // ---- identities: where my candidate REALLY stood in the list ----
        if (unc_in != 0u) hub_order();
        CRIT(9);
        const int truepos = lane < 32 ? hub_pos : F.pos_tail;    // hub_pos = lane when nothing was provisional
        trueslot = (mine && truepos >= 0) ? sig[truepos] : -1;
        wave_sync();
        if (trueslot >= 0) { out_slot->key2[trueslot] = c.key; out_slot->ts2[trueslot] = c.ts; }
        if (!set_out) {                                          // (NaN weights) the kept set itself needed the replay
            provslot = trueslot;
            publish_set(provslot, n_new, new_norm, 0u, 0, 0, 0);
            if (provslot >= 0) out_slot->pos[provslot] = provslot;
            publish_seq(true, true);
            __builtin_amdgcn_s_setprio(0);
        } else {
            if (provslot >= 0) out_slot->pos[provslot] = trueslot;
            publish_seq(false, true);
        }
        CRIT(10);
    }
    c.slot = trueslot;
    if (hub_to_memory) store_row_scatter(h, m, hub, lane, n_new, c, new_norm, tag_base | (unsigned)(wo_h + 1));
    // the new row in dictionary order is the NEXT position's version: that edge's partner task, the hub's readers and
    // the chains of other hubs read it there
    store_row_scatter_at(hub_version(h, m, chain_idx, tpos + 1), k, lane, n_new, c, new_norm, vtag);
    if (lane == 0)                                               // both stages of the incoming slot have been read
        __hip_atomic_store(&in_slot->seq_free, tpos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    HSTAMP(3);
#ifdef ZT_CRIT
    if (lean_done) { CRIT(6); crit_t[7] = 1; }
    if (lane == 0 && mo == 0) atomicAdd((unsigned long long *)&g_crit[8199 * 16 + (lean_done ? 0 : (pre_b.ok ? 6 : (!pre_scale.valid ? 7 : (pre_hash == 2 ? 8 : 9))))], 1ull);
    crit_t[15] = (long long)chain_idx * 100000 + tpos;
    crit_t[12] = mail->t_start;
    if (lane == 0 && mo < 2 && i < 4096 && ((A.B <= 4096 && !A.crit_multi) || A.B >= 12288))   // (model 1 in the upper half: tools/crit_profile.py;
                                                                 //  in a pipelined run the launches over 3+ batches only)
        for (int q = 0; q < 16; ++q) if (q != 14) g_crit[(mo * 4096 + i) * 16 + q] = crit_t[q];
#endif
    (void)wl_fail;
    return true;
}

// The other half of a chain-owned edge (i, model mo), run by a wave of the GENERAL queue on another compute unit: the
// partner's update from the hub's OLD row (utils/util.py:509-564 for the pair (partner, hub)) and the emission of the
// edge's three rows.  The hub's old row is version t of its chain (hub_version: written by the chain in dictionary
// order, hop after hop), the partner's and the negative sample's rows come from memory like everybody's.  The chain
// reads the partner's OLD row as well: its "reads done" flag (cdone) gates the store of the partner's new row.
// Taking this work out of the chain workgroup leaves the wave that holds the chain alone on its SIMD.
__device__ __attribute__((always_inline)) inline void process_chain_partner(const zt_tppr &h, const StreamArgs &A, WaveLds &L, int lane, int i, int mo)
{
    const int k = h.k, B = A.B, n_roles = A.n_roles;
    const int m = A.m_lo + mo;
    const double alpha = h.alpha[m], beta = h.beta[m];
    unsigned *done = h.done + (long long)m * MAX_CHUNK;
    const unsigned *cdone = h.cdone + (long long)m * MAX_CHUNK;
    const unsigned epoch = A.epoch, tag_base = epoch << ORD_BITS, vtag = tag_base | 1u;
    const long long role_stride = A.role_stride;
    const int c = h.owner_of[i];
    const long long hub = h.chain_node[c];
    int my_wo = 0, my_pf = -1, my_hv = -1;
    if (lane < n_roles) { my_wo = h.wo[lane * B + i]; my_pf = h.pflag[lane * B + i]; my_hv = h.hv[lane * B + i]; }
    const int wo_u = __shfl(my_wo, 0), wo_v = __shfl(my_wo, 1), wo_g = __shfl(my_wo, 2);
    const int hv_u = __shfl(my_hv, 0), hv_v = __shfl(my_hv, 1), hv_g = __shfl(my_hv, 2);
    const long long u = A.nodes[i], v = A.nodes[role_stride + i];
    const long long g = n_roles == 3 ? A.nodes[2 * role_stride + i] : u;
    const double tnow = A.tsv[i];
    const long long e = A.eidx[i];
    const bool hub_is_u = u == hub;
    const long long pnode = hub_is_u ? v : u;          // == hub for a self-loop
    const int wo_p = hub_is_u ? wo_v : wo_u;
    const int t = hub_is_u ? wo_u : wo_v;              // the edge's position in the owner's chain = the hub's writer ordinal
    // the partner is a hub whose chain holds this edge too: that chain applies the partner's update, its old row is a version
    const int pchain = pnode != hub ? (hub_is_u ? hv_v : hv_u) : -1;
    // A reader of the partner's row that precedes this edge has not read yet (this task stores that row).  The negative
    // sample's row likewise: a writer waits for the LAST reader before it only, so every reader waits for the reader
    // before it.  (Not for rows held by version: those are never overwritten.)
    {
        const int pf_p = __shfl(my_pf, hub_is_u ? 1 : 0), pf_g = __shfl(my_pf, 2);
        if (pnode != hub && pchain < 0 && pf_p >= 0) (void)wait_flag(done + pf_p, epoch, h.ctl + 2, pf_p);
        if (n_roles == 3 && hv_g < 0 && pf_g >= 0) (void)wait_flag(done + pf_g, epoch, h.ctl + 2, pf_g);
    }
    Row rh, rp, rg;
    STAMP3(4);
    // ---- rows: the partner's and the negative sample's from `rows` (or, hubs, from their chains' versions), the hub's
    // ---- old one from its version slot ----
    const RowSrc psrc = row_src(h, m, pnode, wo_p, pchain, tag_base, vtag), gsrc = row_src(h, m, g, wo_g, hv_g, tag_base, vtag);
    const bool g_own = n_roles == 3 && g != u && g != v;
    unsigned sp = 0, sg = 0;
    const u64 *ver = hub_version(h, m, c, t);
    unsigned sh = load_row_at(ver, k, lane, vtag, rh);
    if (pnode != hub) sp = load_row_at(psrc.base, k, lane, psrc.expect, rp);
    if (g_own) sg = load_row_at(gsrc.base, k, lane, gsrc.expect, rg);
    if (pnode != hub && sp != psrc.expect) (void)load_row_wait_at(psrc.base, k, lane, psrc.expect, rp, h.ctl + 2, (int)pnode, psrc.aux(m), psrc.version);
    if (g_own && sg != gsrc.expect) (void)load_row_wait_at(gsrc.base, k, lane, gsrc.expect, rg, h.ctl + 2, (int)g, gsrc.aux(m), gsrc.version);
    STAMP3(5);
    if (sh != vtag) (void)load_row_wait_at(ver, k, lane, vtag, rh, h.ctl + 2, (int)hub, m | (t << 4), true);      // the chain has not reached this position yet
    // ---- all reads done: later writers of these rows may go ahead ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    st_agent(done + i, epoch);
    STAMP3(6);
    if (pnode != hub && pchain >= 0) {
        // (the partner's chain applies its update; its old row is only emitted here)
    } else if (pnode != hub) {
        // (edge_idx, hub, ts) is the key entering the partner's dictionary
        const u64 nkey = ((u64)(unsigned)e << 32) | (u64)(unsigned)hub;
        Cand cc;
        const int n = merge_pair_reg(L, lane, k, alpha, beta, rp, rh, nkey, tnow, cc);
        const double new_norm = rp.norm * beta + beta;
        (void)wait_flag(cdone + i, epoch, h.ctl + 2, -i - 2);      // the chain has read the partner's old row
        store_row_scatter(h, m, pnode, lane, n, cc, new_norm, tag_base | (unsigned)(wo_p + 1));
        STAMP3(7);
    } else {
        rp = rh;
    }
    if (A.emit) {
        const Row &ru = hub_is_u ? rh : rp, &rv = hub_is_u ? rp : rh;
        if (n_roles == 3 && !g_own) rg = (g == u) ? ru : rv;
        emit_edge(A, k, lane, i, mo, ru, rv, rg, tnow);
    }
#ifdef ZT_CRIT
    if (lane == 0 && mo == 0) atomicMax((unsigned long long *)&g_crit[8191 * 16 + 1], (unsigned long long)__builtin_readcyclecounter());
    if (lane == 0 && mo == 0 && i < 4096) g_crit[i * 16 + 14] = (long long)__builtin_readcyclecounter();   // partner task done
#endif
}

}  // namespace
