// The general hop of k_stream: one (edge, model) task of the in-order queue -- row selection by role, waits on the rows'
// tags, the update of both endpoints, emission of the three output rows; the mailbox structures of the hub chains.
#pragma once

#include "tppr_rows.hpp"

namespace {

// ------------------------------------------------------------ main kernel ----
// LDS mailbox of a chain workgroup: a ring of hand-off slots, one per chain position modulo MAIL_R.  The hub's
// new row passes from the edge at chain position t to the edge at t+1 (held by a sibling wave) through slot
// t % MAIL_R, an LDS round trip instead of a write-through store plus a memory poll -- and in TWO stages:
//   stage 1, the SET in a PROVISIONAL arrangement: the kept entries in ascending order of weight, entries of
//            equal weight in arbitrary order among themselves (`unc` marks them).  This is known after the rank
//            pass (merge_front), before the quicksort replay that decides the order inside such runs;
//   stage 2, the ORDER: pos[s] = dictionary position of the entry at provisional slot s (a permutation inside
//            the runs of equal weight), published after the replay.
// What makes the split pay: numba's argsort only ever COMPARES values, so its dynamics -- and the final slot of
// every list POSITION -- follow from the sequence of values by position, which the provisional arrangement
// already has exactly.  The successor therefore runs its whole update on stage 1 (scales, key matching,
// candidate list, rank pass AND its own replay) and publishes its own stage 1 without waiting for anybody's
// replay; only the identities inside runs of equal weight are settled afterwards, by composing permutations
// along the chain (stage 2: one LDS gather per hop).  The chain's critical path per hop is the front half.
// The one thing that does depend on identities is a key match (or the new key) falling on an entry whose slot
// is still provisional: that hop waits for stage 2 first (process_edge).
// seq_set / seq_ord = chain position + 1 once published (0 at launch).
// (8 slots are enough when the wave that publishes position t also did its preparation; the spine -- tppr_chain.hpp, one
//  wave that runs EVERY critical section with the row in its registers -- is up to a helper's whole cycle ahead of the
//  off-chain halves that release the slots: 16)
constexpr int MAIL_R = 2 * WAVES_PER_WG;
struct MailSlot {
    u64 key[32];
    double ts[32];
    double w[32];
    int pos[32];       // stage 2: dictionary position of the entry at provisional slot s (-1: it is not in the row after all)
    u64 key2[32];      // stage 2: keys / timestamps in dictionary order (the weights by slot are those of stage 1)
    double ts2[32];
    u64 alt_key[32];   // stage 1: the members of a straddling run that were NOT picked (see munc)
    // the header of stage 1 in ONE 16-byte word (one LDS instruction to write, one to read):
    //   norm; meta = len | munc << 8 | n_alt << 16 | sorted << 24; unc
    //   unc   : bit s = the entry at provisional slot s may sit elsewhere in its run of equal weights
    //   munc  : slots [0, munc) hold a PICK of munc members out of a run of munc + n_alt equal weights that straddles
    //           the cut; which members stay is settled by the replay
    //   sorted: the arrangement is ascending by weight (every pruned row; not a row that was never full)
    alignas(16) double norm;
    unsigned meta;
    unsigned unc;
    int seq_set;       // written last of stage 1
    int seq_ord;       // written last of stage 2
    int seq_free;      // = position of the READER once it is done with both stages: the slot may be rewritten
};
typedef unsigned mail_v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mail_hdr_read(const MailSlot *sl, double &norm, int &len, unsigned &unc, int &munc, int &n_alt, int &sorted)
{
    const mail_v4u v = *reinterpret_cast<const mail_v4u *>(&sl->norm);
    norm = __longlong_as_double((long long)(((u64)v.y << 32) | v.x));
    len = (int)(v.z & 0xffu); munc = (int)((v.z >> 8) & 0xffu); n_alt = (int)((v.z >> 16) & 0xffu); sorted = (int)(v.z >> 24);
    unc = v.w;
}
__device__ __forceinline__ void mail_hdr_write(MailSlot *sl, double norm, int len, unsigned unc, int munc, int n_alt, int sorted)
{
    const u64 nb = (u64)__double_as_longlong(norm);
    mail_v4u v;
    v.x = (unsigned)nb; v.y = (unsigned)(nb >> 32);
    v.z = (unsigned)len | ((unsigned)munc << 8) | ((unsigned)n_alt << 16) | ((unsigned)sorted << 24);
    v.w = unc;
    *reinterpret_cast<mail_v4u *>(&sl->norm) = v;
}
// Spine mode (chain_spine): what the helper wave of position t has prepared -- the partner's side of the candidate list,
// sorted, in the helper's own WaveLds; here its uniform part -- and, written back by the spine, what that helper needs for
// the off-chain half.  One record per position modulo PREP_R; a helper holds one position at a time.
constexpr int PREP_R = 16;
struct PrepHdr {
    alignas(16) double norm;   // the norm the scale factors were worked out for (the spine compares it with the row's)
    double scale_s1;
    double norm_next, tnow;
    u64 nkey;
    unsigned meta;             // nb | lenp << 8 | pre_hash << 16 | wave << 20 | ok << 24
    int seq;                   // position + 1 once posted (written last)
    u64 S;                     // spine -> helper: the mask of run starts
    int res;                   // spine -> helper: position + 1 = "the spine ran the section"; -(position + 1) = "yours" (written last)
    int a_seq;                 // duo mode: generation << 12 | position + 1 once the weights wave has left S and the sorted positions (written last)
};
struct Mail {
#ifdef ZT_CRIT
    long long t_start; // core clock when the workgroup started (diagnostic)
#endif
    MailSlot slot[MAIL_R];
    PrepHdr prep[PREP_R];
    int head;          // next position of the chain's edge list
    int a_restart, a_gen;   // duo mode, spine -> weights wave: "everything you have for positions >= a_restart - 1 is void; pick the row up
                            // from the slot position a_restart - 1 publishes" (a_gen written last; it only grows)
};

struct StreamArgs {
    const int *nodes;
    const double *tsv;
    const long long *eidx;
    long long role_stride;
    int B, n_roles, emit, m_lo, n_models;
    int use_chains;    // 0: the grid cannot be guaranteed resident -> every edge goes through the in-order queue
    long long out_rows;
    int *out_nodes, *out_eidx;
    float *out_dt, *out_w;
    unsigned epoch;
    int chain_waves;   // waves of a chain workgroup that take chain hops (the others exit: the chain wave keeps its SIMD)
    int crit_multi;    // diagnostic build: stamps of launches over 3+ batches only (ZT_CRIT_MULTI=1: tools/exp/bench_crit.py)
    int pairs;         // 1: chain waves take two consecutive positions in one critical section where they can (tppr_pair.hpp)
    int sub_B;         // > 0: the launch covers several consecutive batches of sub_B edges (the last may be shorter); the
                       // output rows of batch g form their own [n_models][n_roles][B_g][k] block, blocks back to back
    int *member_done;  // sub_B > 0 and the caller wants batches released one by one: TPPR_MEMBER_WORDS counters (common.hpp), else NULL
};

__device__ __forceinline__ int lds_load_seq(const int *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// spin until *p == want (an LDS word of this workgroup's mailbox); bounded like every other wait
__device__ inline bool wait_seq(const int *p, int want, int *status, int what, int aux, bool hot = false)
{
    unsigned spins = 0;
    long long t0 = 0;
    while (lds_load_seq(p) != want) {
        if (!hot) __builtin_amdgcn_s_sleep(1);           // hot: the next wave on a chain polls back to back
        if ((++spins & 4095u) == 0) {
            const long long now = (long long)wall_clock64();
            if (t0 == 0) t0 = now;
            else if (now - t0 > WAIT_TICKS) { note_timeout(status, 3, what, want, lds_load_seq(p), aux); return false; }
            if (launch_failed(status)) return false;
        }
    }
    asm volatile("" ::: "memory");      // LDS only, in program order behind the load that has just returned (see publish_seq)
    return true;
}

// Apply edge i of the launch for emitted model mo.  mail != nullptr: this wave belongs to the chain
// workgroup of node `hub`; prev_edge = the chain's previous edge (or -1).
// what a chain wave remembers from its previous hop: the hub's norm after it, and the chain position
struct ChainHint {
    double norm_out;
    int tpos;
};

// the three output rows of edge i for emitted model mo (utils/util.py:504-506)
__device__ __forceinline__ void emit_edge(const StreamArgs &A, int k, int lane, int i, int mo, const Row &ru, const Row &rv,
                                          const Row &rg, double tnow)
{
    const int B = A.B, n_roles = A.n_roles;
    long long ou, ov, og;                      // first element of the three output rows of this edge
    if (A.sub_B > 0) {
        const int g = i / A.sub_B, ii = i - g * A.sub_B;
        const int Bg = (B - g * A.sub_B) < A.sub_B ? (B - g * A.sub_B) : A.sub_B;
        const long long base = ((long long)g * A.n_models * n_roles * A.sub_B + (long long)mo * n_roles * Bg) * k;
        ou = base + (long long)ii * k; ov = base + (long long)(Bg + ii) * k; og = base + (long long)(2 * Bg + ii) * k;
    } else {
        const long long ob = (long long)mo * A.out_rows * k;
        ou = ob + (long long)i * k; ov = ob + (A.role_stride + i) * k; og = ob + (2 * A.role_stride + i) * k;
    }
    emit_row(ru, k, lane, tnow, A.out_nodes + ou, A.out_eidx + ou, A.out_dt + ou, A.out_w + ou);
    emit_row(rv, k, lane, tnow, A.out_nodes + ov, A.out_eidx + ov, A.out_dt + ov, A.out_w + ov);
    if (n_roles == 3) emit_row(rg, k, lane, tnow, A.out_nodes + og, A.out_eidx + og, A.out_dt + og, A.out_w + og);
}

// version t of chain c's hub row for model m (see zt_tppr::hubver)
__device__ __forceinline__ u64 *hub_version(const zt_tppr &h, int m, int c, int t)
{
    return h.hubver + (((size_t)m * MAX_CHAINS + c) * (CH_MAX + 1) + t) * h.rg;
}

// the scale factors of chain c's hub at position t for model m (see zt_tppr::hubscale)
__device__ __forceinline__ double *hub_scale(const zt_tppr &h, int m, int c, int t)
{
    return h.hubscale + (((size_t)m * MAX_CHAINS + c) * (CH_MAX + 1) + t) * 4;
}

// Where an access finds its row: in `rows` under the tag of the node's last earlier writer of the launch, or -- a hub's row
// for everybody but the hub's own chain -- in the chain's versions, slot = that writer ordinal, under the launch's version tag.
struct RowSrc {
    const u64 *base;
    unsigned expect;
    bool version;
    bool polled;       // the row must carry `expect` (a row nobody of this launch has written is taken as it is)
    int slot;          // (for a time-out report) the version slot
    __device__ __forceinline__ int aux(int m) const { return version ? (m | (slot << 4)) : m; }
};
__device__ __forceinline__ RowSrc row_src(const zt_tppr &h, int m, long long x, int wo, int hv, unsigned tag_base, unsigned vtag)
{
    RowSrc s;
    s.slot = wo;
    if (hv >= 0) { s.base = hub_version(h, m, hv, wo); s.expect = vtag; s.version = true; s.polled = true; }
    else { s.base = h.rows + ((long long)m * h.N + x) * h.rg; s.expect = wo ? (tag_base | (unsigned)wo) : 0u; s.version = false; s.polled = wo != 0; }
    return s;
}

// (always_inline: with a third call site in k_stream the inliner once left this a real call -- rows in scratch memory, the
//  kernel 17x slower: round 4)
__device__ __attribute__((always_inline)) inline void process_edge(const zt_tppr &h, const StreamArgs &A, WaveLds &L, int lane, int i, int mo, Mail *mail,
                                    long long hub, int prev_edge, int next_edge, int tpos, ChainHint *hint = nullptr,
                                    int chain_idx = -1)
{
    const int k = h.k, B = A.B, n_roles = A.n_roles;
    const int m = A.m_lo + mo;
    const double alpha = h.alpha[m], beta = h.beta[m];
    unsigned *done = h.done + (long long)m * MAX_CHUNK;
    const unsigned epoch = A.epoch, tag_base = epoch << ORD_BITS;
    const unsigned vtag = tag_base | 1u;             // tag of the hub-row versions of this launch
    const long long role_stride = A.role_stride;
#ifdef ZT_CRIT
    long long crit_t[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    STAMP(0);
    HSTAMP(0);
    WL(0, 1); WL(1, blockIdx.x * WAVES_PER_WG + threadIdx.x / WAVE); WL(5, mail ? prev_edge : -2); WL(2, wall_clock64() >> 7);
    int wl_fail = 0;
    unsigned wl_seen = 0;

    // ---- dependencies of this edge's three accesses ----
    int my_wo = 0, my_pf = -1, my_nx = 0, my_hv = -1;
    if (lane < n_roles) {
        my_wo = h.wo[lane * B + i]; my_pf = h.pflag[lane * B + i]; my_nx = h.nxt[lane * B + i];
        if (A.use_chains) my_hv = h.hv[lane * B + i];
    }
    // a reader before me has not read yet.  (Not a chain wave's concern: it stores the hub's row to `rows` once, after the
    // chain's last hop -- nobody else reads that row there in this launch -- and the partner's row not at all.)
    if (mail == nullptr && my_pf >= 0 && !wait_flag(done + my_pf, epoch, h.ctl + 2, my_pf)) wl_fail |= 1;
    wl_fail = __ballot(wl_fail != 0) != 0ull ? 1 : 0;
    WL(0, 2);
    const int wo_u = __shfl(my_wo, 0), wo_v = __shfl(my_wo, 1), wo_g = __shfl(my_wo, 2);
    const int hv_u = __shfl(my_hv, 0), hv_v = __shfl(my_hv, 1), hv_g = __shfl(my_hv, 2);
    // the endpoint with the longer chain still waiting behind it is merged and published first
    const bool v_first = __shfl(my_nx, 1) > __shfl(my_nx, 0);

    const long long u = A.nodes[i], v = A.nodes[role_stride + i];
    const long long g = n_roles == 3 ? A.nodes[2 * role_stride + i] : u;
    const double tnow = A.tsv[i];
    const long long e = A.eidx[i];

    // Does the hub's row reach me through the mailbox?  Yes iff the chain's previous edge is the last
    // writer of the hub before me (the tag it writes is the one I expect).
    bool hub_by_mail = false;
    if (mail != nullptr && prev_edge >= 0) {
        const int prole = A.nodes[prev_edge] == hub ? 0 : 1;
        const int prev_out = h.wo[prole * B + prev_edge] + 1;
        const int mine = (u == hub) ? wo_u : wo_v;
        hub_by_mail = prev_out == mine;
    }

    // Must the hub's new row also go to `rows`?  Only from the chain's LAST hop: every hop's new row goes to the next
    // position's version slot, where everybody else reads it.  This is a correctness rule, not only a saving: the mailbox
    // hand-off is NOT ordered against this wave's row stores, so a successor could otherwise get its (newer) row into
    // `rows` before ours and ours would then overwrite it.
    const bool hub_to_memory = mail == nullptr || next_edge < 0;

    // ---- rows: one memory round trip; poll where a writer of this launch precedes us ----
    Row ru, rv, rg;
    const bool u_mail = hub_by_mail && u == hub, v_mail = hub_by_mail && v == hub && v != u;
    unsigned su = 0, sv = 0, sg = 0;
    // (a chain wave reads its own hub's row from `rows` -- the chain's first hop -- and another hub's by version; a general
    //  task has no in-chain hub as an endpoint, its negative sample may be one)
    const RowSrc src_u = row_src(h, m, u, wo_u, (mail != nullptr && u != hub) ? hv_u : -1, tag_base, vtag);
    const RowSrc src_v = row_src(h, m, v, wo_v, (mail != nullptr && v != hub) ? hv_v : -1, tag_base, vtag);
    const RowSrc src_g = row_src(h, m, g, wo_g, hv_g, tag_base, vtag);
    if (!u_mail) su = load_row_at(src_u.base, k, lane, src_u.expect, ru);
    if (v != u && !v_mail) sv = load_row_at(src_v.base, k, lane, src_v.expect, rv);
    // (a chain wave applies the HUB's update only: the partner's update and the emission of this edge's rows are a
    //  general task of their own, process_chain_partner -- the negative sample's row is not needed here)
    const bool g_own = mail == nullptr && n_roles == 3 && g != u && g != v;
    if (g_own) sg = load_row_at(src_g.base, k, lane, src_g.expect, rg);
    WL(0, 3);
    if (!u_mail && src_u.polled && su != src_u.expect)
        if (!load_row_wait_at(src_u.base, k, lane, src_u.expect, ru, h.ctl + 2, (int)u, src_u.aux(m), src_u.version, &wl_seen)) wl_fail |= 2;
    WL(0, 4);
    if (v != u && !v_mail && src_v.polled && sv != src_v.expect)
        if (!load_row_wait_at(src_v.base, k, lane, src_v.expect, rv, h.ctl + 2, (int)v, src_v.aux(m), src_v.version, &wl_seen)) wl_fail |= 4;
    WL(0, 5);
    if (g_own && src_g.polled && sg != src_g.expect)
        if (!load_row_wait_at(src_g.base, k, lane, src_g.expect, rg, h.ctl + 2, (int)g, src_g.aux(m), src_g.version, &wl_seen)) wl_fail |= 8;
    int pre_hash = 0;                           // 1: partner entered into this wave's hash table, 2: with a clash
    const bool sw = v_first && u != v;          // v's new row is computed and published first
    MailSlot *in_slot = hub_by_mail ? &mail->slot[(tpos - 1) % MAIL_R] : nullptr;
    MailSlot *out_slot = mail != nullptr ? &mail->slot[tpos % MAIL_R] : nullptr;
    // the hub's row arrives in set order and its order later (two-stage hand-off); otherwise rows are in
    // dictionary order
    bool hub_ordered = true;
    unsigned hub_unc = 0u;                      // slots of the hub's row that are provisional (stage 2 pending)
    int hub_munc = 0, hub_nalt = 0;             // slots [0, hub_munc) hold a pick out of a straddling run; its other members
    u64 hub_alt = 0ull;                         // (this lane's, if lane < hub_nalt)
    bool hub_final = true;                      // stage 1 was already the dictionary order
    const bool hub_is_u = u == hub;
    PreScale pre_scale;
    pre_scale.valid = false;
    PreB pre_b;
    pre_b.ok = false;
    int hub_sorted = 0;                         // the hub's row arrived ascending by weight
    int free_seen = -1;                         // seq_free of my ring slot as read with the row (-1: not read)
    if (hub_by_mail) {
        // everything else is in registers by now; the hub's row arrives through LDS
        WL(0, 6);
        // while waiting: the partner of the first merge goes into the hash table already (merge_front, pre)
        {
            const long long x1_0 = sw ? v : u;
            const Row &rp = sw ? ru : rv;
            const int lenp = (rp.norm != 0.0) ? rp.len : 0;
            if (x1_0 == hub && u != v && lenp > 0) {
                const int h2 = key_hash(rp.key);
                if (lane < lenp) L.htab[h2] = lane;
                wave_sync();
                const int back = lane < lenp ? L.htab[h2] : lane;
                const bool clash = __ballot(lane < lenp && back != lane) != 0ull;
                if (clash && lane < lenp && back == lane) L.htab[h2] = -1;
                pre_hash = clash ? 2 : 1;
                wave_sync();
            }
        }
        // ... and the scale factors for the norm the hub will have if the hops since my last one were ordinary
        if (hint != nullptr && hint->tpos >= 0 && tpos - hint->tpos <= 16) {
            double pn = hint->norm_out;
            for (int q = hint->tpos + 1; q < tpos; ++q) pn = pn * beta + beta;
            if (pn != 0.0) {
                const double nn = pn * beta + beta;
                pre_scale.norm = pn;
                pre_scale.scale_s1 = pn / nn * beta;
                pre_scale.scale_s2 = beta / nn * (1.0 - alpha);
                pre_scale.valid = true;
            }
        }
        // ... and the partner's side of the candidate list, sorted (merge_front_fast)
        {
            const long long x1_0 = sw ? v : u, x2_0 = sw ? u : v;
            if (x1_0 == hub && u != v && pre_hash != 2)
                prepare_b(lane, k, alpha, sw ? ru : rv, ((u64)(unsigned)e << 32) | (u64)(unsigned)x2_0, tnow, pre_scale, pre_b,
                          key_hash((sw ? ru : rv).key));
        }
        // All rows that come from memory have arrived (the hub's comes through LDS): "reads done" can be said now
        // instead of on the chain (see below)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        st_agent(h.cdone + (long long)m * MAX_CHUNK + i, epoch);
        // From here to the publication of the new kept set this wave IS the chain: it shares its SIMD with a wave that
        // is busy with the off-chain half of an earlier hop (replay, partner's update, emission), and at equal
        // priority the two alternate issue slots.
        // (Only once the row is there: a wave that SPINS at high priority starves the off-chain work of its SIMD
        // mate, and later hops wait for that work's results.)
        // (Poll and read as ONE batch of LDS instructions -- sequence word first, in-order execution makes that safe --
        // was measured: the seven waiting waves then issue ten LDS reads per poll, and the hop gets 6 % slower.)
        // Waves whose turn is two or more hops away doze (the hop before their predecessor's has not been published):
        // seven waves polling every ~200 cycles take LDS and issue slots from the one that works.
        if (tpos >= 2) {
            const int *far = &mail->slot[(tpos - 2) % MAIL_R].seq_set;
            unsigned spins = 0;
            while (lds_load_seq(far) != tpos - 1 && lds_load_seq(&in_slot->seq_set) != tpos) {
                __builtin_amdgcn_s_sleep(8);
                if ((++spins & 1023u) == 0 && launch_failed(h.ctl + 2)) break;
            }
        }
        if (!wait_seq(&in_slot->seq_set, tpos, h.ctl + 2, i, prev_edge)) wl_fail |= 16;
        __builtin_amdgcn_s_setprio(3);
        CRIT(0);
#ifdef ZT_STAMP
        { const int g_stamp_i = mo == 0 ? i : -1; STAMP2(6); }
#endif
        // one batch of LDS reads: the row, its provisional marks, and whether my own ring slot is free again
        Row rm;
        mail_hdr_read(in_slot, rm.norm, rm.len, hub_unc, hub_munc, hub_nalt, hub_sorted);
        rm.key = in_slot->key[lane & 31]; rm.ts = in_slot->ts[lane & 31]; rm.w = in_slot->w[lane & 31];
        hub_alt = in_slot->alt_key[lane & 31];
        free_seen = lds_load_seq(&out_slot->seq_free);
        if (hub_is_u) ru = rm; else rv = rm;
        hub_ordered = hub_unc == 0u && hub_munc == 0;   // nothing provisional: the arrangement is the dictionary order
        hub_final = hub_ordered;
#ifdef ZT_CRIT
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        CRIT(8);
    }
    // stage 2 of the hub's row: dictionary position of my set-order entry (identity when the row came from memory)
    int hub_pos = lane;
    auto hub_order = [&]() {
        if (hub_ordered) return;
        if (!wait_seq(&in_slot->seq_ord, tpos, h.ctl + 2, i, -prev_edge - 2)) wl_fail |= 32;
        hub_pos = in_slot->pos[lane & 31];
        hub_ordered = true;
    };
    // the hub's old row in dictionary order (what the partner's update, a self-loop and emission read): the
    // keys of stage 2; the weights by slot are the same in both arrangements
    auto hub_to_dict = [&]() {
        hub_order();
        if (hub_final) return;
        Row &r = hub_is_u ? ru : rv;
        r.key = in_slot->key2[lane & 31]; r.ts = in_slot->ts2[lane & 31];
        hub_pos = lane;
        hub_final = true;
        hub_unc = 0u; hub_munc = 0; hub_nalt = 0;
    };
    // the split hand-off applies when the hub's update is the first of the two: it can then run ahead of the order
    const bool split = hub_by_mail && u != v && (sw ? v : u) == hub;
    // the previous position's slot is mine to release, whether or not the row came through it
    auto release_in = [&]() {
        if (mail != nullptr && tpos >= 1 && lane == 0)
            __hip_atomic_store(&mail->slot[(tpos - 1) % MAIL_R].seq_free, tpos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    if (hub_by_mail && !split) hub_to_dict();
    if (!split) release_in();
    if (v == u) rv = ru;

    STAMP(1);
    HSTAMP(1); HSTAMP(2);
    WL(0, 7); WL(3, wall_clock64() >> 7); WL(6, wl_fail | (hub_by_mail ? 256 : 0)); if (wl_fail & 14) WL(7, wl_seen);
    // ---- all reads done: later writers of these rows may go ahead ----
    // The row loads above must have RETURNED before a later writer may see the flag (the row of a negative
    // sample is not consumed until emission, so nothing else orders its loads): drain vmcnt explicitly.  A
    // release store at agent scope would do it too, but it also writes the XCD's L2 back (buffer_wbl2) on
    // every hop; the rows themselves travel as write-through sc1 granules and need no such flush.
    if (!hub_by_mail) {                                               // (a hop whose hub row comes by mail has said so already)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (mail == nullptr) st_agent(done + i, epoch);              // every lane, same word (no lane-0 branch, see the dequeue)
        else st_agent(h.cdone + (long long)m * MAX_CHUNK + i, epoch);   // the chain's reads: the partner task may store the partner's new row
    }

    // ---- both directions from the OLD rows (utils/util.py:509-564); each new row is
    // ---- written back (utils/util.py:567-574) as soon as it exists: the tagged row IS the hand-off
    {
        const Row &r1 = sw ? rv : ru, &r2 = sw ? ru : rv;
        const long long x1 = sw ? v : u, x2 = sw ? u : v;
        const int o1 = sw ? wo_v : wo_u, o2 = sw ? wo_u : wo_v;
        // (edge_idx, s2, ts) is the key entering s1's dictionary
        const bool reg_path = k <= REG_K_MAX;   // 2k+1 candidates fit one wavefront: register-resident merge
        Cand c;
        // a slot of the ring is reused every MAIL_R positions: wait until the reader of its previous content
        // (chain position tpos - MAIL_R + 1) has let go of it
        auto ring_free = [&]() {
            if (tpos >= MAIL_R) {
                if (free_seen == tpos - MAIL_R + 1) asm volatile("" ::: "memory");
                else if (!wait_seq(&out_slot->seq_free, tpos - MAIL_R + 1, h.ctl + 2, i, -1)) wl_fail |= 64;
            }
        };
        auto publish_set = [&](int sidx, int n, double new_norm, unsigned unc = 0u, int munc = 0, int n_alt = 0, int sorted = 0) {
            if (sidx >= 0) { out_slot->key[sidx] = c.key; out_slot->ts[sidx] = c.ts; out_slot->w[sidx] = c.w; }
            if (lane == 0) mail_hdr_write(out_slot, new_norm, n, unc, munc, n_alt, sorted);
        };
        auto publish_seq = [&](bool set, bool ord) {
            // The mailbox lives in LDS and a wave's LDS instructions execute in program order: the sequence word, issued
            // after the data, becomes visible after it -- no wait.  (A workgroup-scope release fence would also drain this
            // wave's global stores, vmcnt(0), and wait for the LDS writes to finish: ~100 cycles on the chain.)
            asm volatile("" ::: "memory");
            if (lane == 0 && set) __hip_atomic_store(&out_slot->seq_set, tpos + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (lane == 0 && ord) __hip_atomic_store(&out_slot->seq_ord, tpos + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        // one full pair update, rows in dictionary order; the hub's new row also goes to the mailbox (both stages)
        auto update = [&](const Row &a, const Row &b, long long xa, long long xb, int oa, int pre, int stamp) {
            const u64 nkey = ((u64)(unsigned)e << 32) | (u64)(unsigned)xb;
            int n;
            if (reg_path) {
                n = merge_pair_reg(L, lane, k, alpha, beta, a, b, nkey, tnow, c, pre, stamp);
            } else {
                n = merge_pair(L, lane, k, alpha, beta, a, b, nkey, tnow, c.key, c.ts, c.w, pre, stamp);
                c.slot = lane < n ? lane : -1;
            }
            const double new_norm = a.norm * beta + beta;
            if (mail != nullptr && xa == hub) {
                if (hint != nullptr) { hint->norm_out = new_norm; hint->tpos = tpos; }
                ring_free();
                publish_set(c.slot, n, new_norm);
                if (c.slot >= 0) out_slot->pos[c.slot] = c.slot;
                publish_seq(true, true);
                HSTAMP(4);
                __builtin_amdgcn_s_setprio(0);
            }
            if (hub_to_memory || xa != hub) store_row_scatter(h, m, xa, lane, n, c, new_norm, tag_base | (unsigned)(oa + 1));
            if (mail != nullptr && xa == hub) store_row_scatter_at(hub_version(h, m, chain_idx, tpos + 1), k, lane, n, c, new_norm, vtag);
        };
        if (!split && mail != nullptr) {
            // a chain wave whose hub row did not come through the mailbox in set order (first hop, self-loop, the partner is
            // the busier node): the hub's update in one piece (version 0, the row the chain's first hop finds in `rows`, was
            // stored when the workgroup started: k_stream)
            if (x1 == hub) update(r1, r2, x1, x2, o1, pre_hash, mo == 0 ? i : -1);
            else update(r2, r1, x2, x1, o2, 0, -1);
            STAMP(2);
            WL(0, 8);
        } else if (!split) {
            update(r1, r2, x1, x2, o1, pre_hash, mo == 0 ? i : -1);
            STAMP(2);
            WL(0, 8);
            if (u != v) update(r2, r1, x2, x1, o2, 0, -1);
        } else {
            // ---- the hub's update on a row that may still be in its provisional arrangement ----
            CRIT(9);
            const u64 nkey = ((u64)(unsigned)e << 32) | (u64)(unsigned)x2;
            const double new_norm = r1.norm * beta + beta;
            if (hint != nullptr) { hint->norm_out = new_norm; hint->tpos = tpos; }
            Front F;
            bool settled = hub_final;                                    // the row is known to be the dictionary
            if (!settled && hub_munc > 0) {
                // members of a straddling run that were not picked may turn out to be in the row: a key match (any
                // occupied hash slot counts) or the new key falling on one of them needs the real row
                bool t = lane < hub_nalt && hub_alt == nkey;
                if (pre_hash == 1) t = t || (lane < hub_nalt && L.htab[key_hash(hub_alt)] >= 0);
                if (__ballot(t) != 0ull || pre_hash == 2) { hub_to_dict(); settled = true; pre_hash = pre_hash == 2 ? 2 : 1; }
            }
            CRIT(10);
            if (!(hub_sorted && merge_front_fast(L, lane, k, r1, r2, pre_b, pre_scale, nkey, tnow, pre_hash == 1 || pre_b.len2 == 0, false, F CRIT_PASS)))
                merge_front(L, lane, k, alpha, beta, r1, r2, nkey, tnow, F, pre_hash, mo == 0 ? i : -1, &pre_scale CRIT_PASS);
            CRIT(1);
            if (!settled) {
                // a key match (or the new key) on an entry whose slot is provisional: the weights by position would
                // depend on identities.  Likewise a picked member of a straddling run that would be kept: whether it
                // is in the row at all is not known yet.  Then: settle the row first and start over.
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
                    merge_front(L, lane, k, alpha, beta, r1, r2, nkey, tnow, F, 0, -1);
#ifdef ZT_STAMP
                    if (lane == 0) atomicAdd(&g_paths[6], 1);
#endif
                }
            }
#ifdef ZT_STAMP
            if (lane == 0 && mo == 0) atomicAdd(&g_paths[7], 1);                        // split hops of model 0
            if (lane == 0 && mo == 0 && F.mode >= FR_STRADDLE) atomicAdd(&g_paths[3], 1);   // ... whose kept set needs the replay
            if (lane == 0 && mo == 0 && F.mode == FR_TIES) atomicAdd(&g_paths[2], 1);
            if (lane == 0 && mo == 0 && F.mode == FR_RANKS) atomicAdd(&g_paths[1], 1);
#endif
            const unsigned unc_in = settled ? 0u : hub_unc;              // provisional slots of the row as I used it
            c.key = F.key; c.ts = F.ts; c.w = F.w;
            const bool mine = (F.live >> lane) & 1ull;
            const int pos_prov = lane < 32 ? lane : F.pos_tail;          // my candidate's place in the list as it arrived
            const int drop = F.n - k;
            int n_new = F.n <= k ? F.n : k, provslot = -1, trueslot = -1;
            unsigned unc_out = 0u;
            int munc_out = 0, nalt_out = 0;
            bool set_out = false, final_out = false;
            ring_free();
            if (F.mode == FR_NOPRUNE) {
                provslot = mine ? pos_prov : -1;                         // s1's entries keep their slots, and their doubts
                unc_out = unc_in;
                set_out = true;
            } else if (F.mode == FR_RANKS) {
                provslot = F.keep ? F.lt - drop : -1;                    // all kept weights distinct: nothing provisional
                set_out = true;
            } else if (F.mode == FR_TIES || F.mode == FR_STRADDLE) {
                // ascending by weight; equal weights take the slots of their run in lane order.
                // A run that STRADDLES the cut (g members of which j stay): the first j by lane are picked for slots
                // [0, j) -- the run has the smallest kept weight -- and the others go along as alternates.
                int ltG = -1, j = 0;
                u64 Gm = 0ull;
                if (F.mode == FR_STRADDLE) {
                    ltG = wave_max0(mine && F.lt < drop ? F.lt + 1 : 0) - 1;
                    Gm = __ballot(mine && F.lt == ltG);
                    j = ltG + __popcll(Gm) - drop;
                }
                const bool certain = mine && F.lt >= drop;
                // a run of g equal weights at rank r claims bit r only (rank_pass): its members are the certain
                // candidates whose next rank is unclaimed.  Run by run (there are two or three), the members take
                // consecutive slots in lane order -- registers only.
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
#ifdef ZT_STAMP
            { const int g_stamp_i = mo == 0 ? i : -1; STAMP2(7); }
#endif
            STAMP(2);
            WL(0, 8);
            if (!final_out) {
                // ---- my own replay: final slot of every list POSITION (identity-free, see Mail) ----
                const int slot_c = merge_order(L, lane, k, F, lane, &n_new, mo == 0 ? i : -1);
                int *sig = L.sel;                                        // final slot by list position
                if (mine) sig[pos_prov] = slot_c;
                wave_sync();
                // ---- identities: where my candidate REALLY stood in the list ----
                if (unc_in != 0u) hub_order();
                const int truepos = lane < 32 ? hub_pos : F.pos_tail;    // hub_pos = lane when nothing was provisional
                trueslot = (mine && truepos >= 0) ? sig[truepos] : -1;
                wave_sync();
                if (trueslot >= 0) { out_slot->key2[trueslot] = c.key; out_slot->ts2[trueslot] = c.ts; }
                if (!set_out) {                                          // (NaN weights) the kept set itself needed the replay
                    provslot = trueslot;
                    publish_set(provslot, n_new, new_norm, 0u);
                    if (provslot >= 0) out_slot->pos[provslot] = provslot;
                    publish_seq(true, true);
                    __builtin_amdgcn_s_setprio(0);
                } else {
                    if (provslot >= 0) out_slot->pos[provslot] = trueslot;
                    publish_seq(false, true);
                }
            }
            c.slot = trueslot;
            if (hub_to_memory) store_row_scatter(h, m, x1, lane, n_new, c, new_norm, tag_base | (unsigned)(o1 + 1));
            // the new row in dictionary order is the NEXT position's version: its partner task reads it there
            store_row_scatter_at(hub_version(h, m, chain_idx, tpos + 1), k, lane, n_new, c, new_norm, vtag);
            release_in();
        }
    }
    if (n_roles == 3 && !g_own) rg = (g == u) ? ru : rv;

    // ---- emission is off the critical path (utils/util.py:504-506) ----
    // (Leaving the emission of hub edges out -- as if other compute units did it -- does not make the chain faster.)
    if (A.emit && mail == nullptr) emit_edge(A, k, lane, i, mo, ru, rv, rg, tnow);
#ifdef ZT_CRIT
    if (lane == 0 && mo == 0 && mail != nullptr && i < 8192)
        for (int q = 0; q < 16; ++q) g_crit[i * 16 + q] = crit_t[q];
#endif
    STAMP(3);
    HSTAMP(3);
    WL(4, wall_clock64() >> 7); WL(0, 9);
    (void)wl_fail;
}


}  // namespace
