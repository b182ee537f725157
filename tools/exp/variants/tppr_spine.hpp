// Spine / duo mode of the hub chains (round 5): built, bit-exact, measured SLOWER than the mailbox hop,
// and therefore not part of the product library: this header is compiled only into variant builds
//     tools/build_variant.sh chainvariants "-DZT_CHAIN_VARIANTS -I/root/repo/tools/exp/variants"
// (tppr_chain.hpp includes it inside its anonymous namespace; tests/test_tppr_gpu.py runs its tests against such a build:
//  ZT_TEST_LIB=tools/out/chainvariants/libzebra_amd.so).
#pragma once

// ---------------------------------------------------------------------------------------------------------------------
// Spine mode (round 5).  With the mailbox alone a hub's row changes waves at every hop: the wave that publishes position t
// writes the kept set to LDS, the next one polls, reads header and row back, unpacks -- ~1 000 of the section's 2 300 clocks
// are that hand-off, and it cannot overlap with anything because the row IS the dependency.  In spine mode ONE wave of the
// chain workgroup (the spine) runs every critical section and keeps the row in its registers from hop to hop; the other
// seven (helpers) do what does not depend on the hub's row -- the partner's row from memory, its hash table, its scaled
// and sorted side of the candidate list (prepare_b) -- post that in LDS (spine_post: per-lane part in the helper's own
// WaveLds, uniform part in Mail::prep), and take the hop's off-chain half (replay, order hand-off, version store) from what
// the spine hands back: the mask of run starts and every candidate's (rank, sorted position).  The mailbox ring is still
// written, by the spine, exactly as a lean hop writes it: it is what the helpers' off-chain halves and the order hand-off
// read, and it is the fallback -- whenever a precondition of the lean section fails (or the helper could not prepare) the
// spine answers "yours", the helper runs the hop the old way from the mailbox (chain_hop / process_edge unchanged), and the
// spine picks the row up from the slot that hop publishes.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void spine_post(PrepHdr *P, WaveLds &L, int lane, int tpos, const PreB &B, const PreScale &ps,
                                           u64 nkey, double tnow, int lenp, int pre_hash)
{
    const bool ok = B.ok;
    if (ok && lane >= 32) {
        L.key[lane - 32] = B.cb_key; L.ts[lane - 32] = B.cb_ts; L.w[lane - 32] = B.cb_w;
        L.key[32 + lane] = (u64)__double_as_longlong(B.sw);
        L.sel[lane - 32] = B.sid;
    }
    if (lane == 0) {
        mail_v4u a, b, c;
        const u64 n0 = (u64)__double_as_longlong(ps.norm), s1 = (u64)__double_as_longlong(ps.scale_s1);
        const u64 n1 = (u64)__double_as_longlong(ps.norm_next), tn = (u64)__double_as_longlong(tnow);
        a.x = (unsigned)n0; a.y = (unsigned)(n0 >> 32); a.z = (unsigned)s1; a.w = (unsigned)(s1 >> 32);
        b.x = (unsigned)n1; b.y = (unsigned)(n1 >> 32); b.z = (unsigned)tn; b.w = (unsigned)(tn >> 32);
        c.x = (unsigned)nkey; c.y = (unsigned)(nkey >> 32);
        c.z = ok ? ((unsigned)B.nb | ((unsigned)lenp << 8) | ((unsigned)pre_hash << 16) | ((unsigned)(threadIdx.x / WAVE) << 20) | (1u << 24)) : 0u;
        c.w = 0u;
        __hip_atomic_store(&P->res, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        *reinterpret_cast<mail_v4u *>(&P->norm) = a;
        *reinterpret_cast<mail_v4u *>(&P->norm_next) = b;
        P->nkey = nkey;
        P->meta = c.z;
        asm volatile("" ::: "memory");                   // (LDS, one wave: program order)
        __hip_atomic_store(&P->seq, tpos + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// a position chain_hop does not take (the chain's first hop, a self-loop): "not prepared" -- the spine leaves it to this wave
__device__ __forceinline__ void spine_post_none(Mail *mail, int lane, int tpos)
{
    PrepHdr *P = &mail->prep[tpos % PREP_R];
    if (lane == 0) {
        __hip_atomic_store(&P->res, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        P->meta = 0u;
        asm volatile("" ::: "memory");
        __hip_atomic_store(&P->seq, tpos + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// the spine's word for position tpos: tpos + 1 (it ran the section) or -(tpos + 1) (this wave's); 0 after a time-out
__device__ inline int spine_wait_res(const int *p, int tpos, int *status, int what)
{
    unsigned spins = 0;
    long long t0 = 0;
    for (;;) {
        const int r = lds_load_seq(p);
        if (r == tpos + 1 || r == -(tpos + 1)) { asm volatile("" ::: "memory"); return r; }
        __builtin_amdgcn_s_sleep(2);
        if ((++spins & 4095u) == 0) {
            const long long now = (long long)wall_clock64();
            if (t0 == 0) t0 = now;
            else if (now - t0 > WAIT_TICKS) { note_timeout(status, 3, what, tpos + 1, r, -7); return 0; }
            if (launch_failed(status)) return 0;
        }
    }
}

__device__ __forceinline__ u64 push_u64(u64 v, int dst)
{
    const unsigned lo = (unsigned)push_i32((int)(unsigned)v, dst), hi = (unsigned)push_i32((int)(unsigned)(v >> 32), dst);
    return ((u64)hi << 32) | lo;
}

// a prepared side as the spine fetches it ahead of time: the uniform part (sequence word first: LDS executes a wave's
// instructions in order, so data read behind a sequence word that says "posted" is the posted data) and the per-lane part
// (lanes >= 32; from the WaveLds of the wave named in meta)
struct SpSide {
    int seq;
    mail_v4u a, b;
    u64 nkey;
    unsigned meta;
    u64 key;
    double ts, w, sw;
    int sid;
    bool lanes;        // the per-lane part has been asked for
};
__device__ __forceinline__ void sp_hdr_load(const PrepHdr *P, SpSide &H)
{
    H.seq = lds_load_seq(&P->seq);
    H.a = *reinterpret_cast<const mail_v4u *>(&P->norm);
    H.b = *reinterpret_cast<const mail_v4u *>(&P->norm_next);
    H.nkey = P->nkey;
    H.meta = P->meta;
    H.lanes = false;
}
__device__ __forceinline__ void sp_lanes_load(const WaveLds &Lw, int idx, SpSide &H)
{
    H.key = Lw.key[idx]; H.ts = Lw.ts[idx]; H.w = Lw.w[idx];
    H.sw = __longlong_as_double((long long)Lw.key[64 + idx]);
    H.sid = Lw.sel[idx];
    H.lanes = true;
}

// Written for instruction count AND for latency: one wave issues an instruction every ~5 clocks and an LDS round trip is
// ~100, so (a) every test of a hop is folded into ONE uniform flag and ONE branch in front of the first write; (b) the
// recurrence from hop to hop is the WEIGHTS alone -- sorted by the network, the kept ones are the sorted lanes from `drop`
// up, one lane shift (ds_bpermute) puts them where the next network wants them -- keys and time stamps follow through the
// (sorted position -> home lane) push while the scalar unit works out the cut; (c) the next position's prepared side is
// fetched while this one's network runs (header) and right behind it (per-lane part); (d) the loop is unrolled by two so
// that "this side" and "the next" need no moves.  A hop that fails a test has clobbered the weights: the spine picks the row
// up from the mailbox after the helper's hop anyway.
template <bool DUO>
__device__ __attribute__((always_inline)) inline void chain_spine(const zt_tppr &h, WaveLds *lds, int lane, Mail *mail, int len)
{
    int duo_gen = 0;                                                            // (DUO) generation of the weights wave's records that are valid
    const int k = h.k;
    const unsigned kmask = (1u << k) - 1u;
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    const bool low = __builtin_amdgcn_inverse_ballot_w64(0xffffffffull);        // lanes 0..31: the hub's entries
    const int idx = lane & 31;
    int *status = h.ctl + 2;
#define ZT_U(x) __builtin_amdgcn_readfirstlane((int)(x))
    // the hub's row, provisional arrangement (ascending by weight once pruned): entry s in lane s
    u64 hkey = 0ull;
    double hts = 0.0, hw = 0.0;
    unsigned h0 = 0u, h1 = 0u;                                                  // (norm: bit pattern)
    int n1 = 0, munc = 0, nalt = 0, sorted = 0;
    bool have = false;
    int n_ran = 0;
    SpSide HA, HB;
    HA.seq = 0; HB.seq = 0; HA.lanes = false; HB.lanes = false;
    auto hop = [&](const int t, SpSide &Hc, SpSide &Hn) -> bool {
        PrepHdr *P = &mail->prep[t % PREP_R];
        MailSlot *out_slot = &mail->slot[t % MAIL_R], *in_slot = &mail->slot[(t + MAIL_R - 1) % MAIL_R];
        if (__builtin_expect(!have && t > 0, 0)) {
            // (after a hop that was somebody else's) the row as that hop published it
            __builtin_amdgcn_s_setprio(1);
            if (!wait_seq(&in_slot->seq_set, t, status, -3, t)) return false;
            __builtin_amdgcn_s_setprio(3);
            double hn; int a0, a1, a2, a3; unsigned a4;
            mail_hdr_read(in_slot, hn, a0, a4, a1, a2, a3);
            hkey = in_slot->key[idx]; hts = in_slot->ts[idx]; hw = in_slot->w[idx];
            const u64 nbits = (u64)__double_as_longlong(hn);
            h0 = (unsigned)ZT_U((unsigned)nbits); h1 = (unsigned)ZT_U((unsigned)(nbits >> 32));
            n1 = ZT_U(a0); munc = ZT_U(a1); nalt = ZT_U(a2); sorted = ZT_U(a3);
            have = true;
        }
        if (__builtin_expect(ZT_U(Hc.seq) != t + 1, 0)) {
            __builtin_amdgcn_s_setprio(1);
            if (!wait_seq(&P->seq, t + 1, status, -4, t)) return false;
            __builtin_amdgcn_s_setprio(3);
            sp_hdr_load(P, Hc);
        }
        const unsigned meta = (unsigned)ZT_U(Hc.meta);
        bool go = have && ((meta >> 24) & 1u) != 0u;
        Hn.seq = 0;
        if (go) {
            const int nb = (int)(meta & 0xffu), lenp = (int)((meta >> 8) & 0xffu), pre_hash = (int)((meta >> 16) & 0xfu);
            WaveLds &Lw = lds[(meta >> 20) & 0xfu];
            if (__builtin_expect(!Hc.lanes, 0)) sp_lanes_load(Lw, idx, Hc);
            if (t + 1 < len) sp_hdr_load(&mail->prep[(t + 1) % PREP_R], Hn);   // (looked at after the network)
            const int fs = lds_load_seq(&out_slot->seq_free);
            const int n = n1 + nb, drop_ = n - k;
            bool fail = sorted == 0 || h0 != (unsigned)ZT_U(Hc.a.x) || h1 != (unsigned)ZT_U(Hc.a.y) || (h0 | h1) == 0u || n1 <= 0 || drop_ <= 0 || n > 63;
            const int drop = fail ? 1 : drop_;
            const double scale_s1 = __longlong_as_double((long long)(((u64)Hc.a.w << 32) | Hc.a.z));
            const double tnow = __longlong_as_double((long long)(((u64)Hc.b.w << 32) | Hc.b.z));
            const u64 nkey = Hc.nkey;
            const bool table = lenp > 0;
            const bool in1 = lane < n1;
            u64 ckey = Hc.key;
            double cts = Hc.ts, cw = Hc.w, sw = Hc.sw;
            int sid = Hc.sid;
            if (low) { ckey = hkey; cts = hts; cw = hw * scale_s1; sw = in1 ? cw : inf; sid = lane; }
            const unsigned ma = pre_hash == 5 ? 0x27D4EB2Fu : (pre_hash == 3 ? 0x85EBCA77u : 0x9E3779B1u);
            const unsigned mb = pre_hash == 5 ? 0x165667B1u : (pre_hash == 3 ? 0xC2B2AE3Du : 0x85EBCA77u);
            if (__builtin_expect(munc > 0, 0)) {
                // members of the previous hop's straddling run that were not picked may turn out to be in the row
                const u64 alt = in_slot->alt_key[idx];
                bool x = lane < nalt && alt == nkey;
                if (table) x = x || (lane < nalt && Lw.htab[(int)((((unsigned)alt * ma) ^ ((unsigned)(alt >> 32) * mb)) >> 22)] >= 0);
                fail = fail || __ballot(x) != 0ull;
            }
            // is a key of the hub's row in the partner's row?  Read now, looked at after the network
            const int cand = (table && in1) ? Lw.htab[(int)((((unsigned)ckey * ma) ^ ((unsigned)(ckey >> 32) * mb)) >> 22)] : -1;
            int sp;
            u64 S;
            if (DUO) {
                // the weights wave (chain_weights) has run the network on the row's weights: sorted positions + run starts from
                // its record -- valid iff it carries this generation (nothing since its last pick-up of the row was left to a helper).
                // (A position that has failed a test already is not waited for: the weights wave may never produce it.)
                sp = 0; S = 1ull;
                if (!fail) {
                    const int want = (duo_gen << 12) | (t + 1);
                    unsigned spins = 0;
                    while (lds_load_seq(&P->a_seq) != want) {
                        if ((++spins & 8191u) == 0 && launch_failed(status)) return false;
                    }
                    asm volatile("" ::: "memory");
                    sp = Lw.sort.r[lane];
                    const u64 sv = P->S;
                    S = ((u64)(unsigned)ZT_U((unsigned)(sv >> 32)) << 32) | (unsigned)ZT_U((unsigned)sv);
                }
            } else {
            merge_stage<32>(sw, sid);
            merge_stage<16>(sw, sid);
            merge_stage<8>(sw, sid);
            merge_stage<4>(sw, sid);
            merge_stage<2>(sw, sid);
            merge_stage<1>(sw, sid);
            // ---- the next row's weights: sorted lane j + drop -> lane j (lanes >= k: whatever; n1 = k masks them) ----
            {
                const long long swb = __double_as_longlong(sw);
                const int from = (lane + drop) << 2;
                const unsigned wl = (unsigned)__builtin_amdgcn_ds_bpermute(from, (int)(unsigned)(swb & 0xffffffffll));
                const unsigned wh = (unsigned)__builtin_amdgcn_ds_bpermute(from, (int)(swb >> 32));
                hw = __longlong_as_double((long long)(((u64)wh << 32) | wl));
            }
            sp = push_i32(lane, sid);                            // sorted position to the candidate's lane
            const long long swb = __double_as_longlong(sw);
            const int llo = __builtin_amdgcn_mov_dpp((int)(unsigned)(swb & 0xffffffffll), 0x138, 0xf, 0xf, true);   // wave_shr:1
            const int lhi = __builtin_amdgcn_mov_dpp((int)(swb >> 32), 0x138, 0xf, 0xf, true);
            S = __ballot(__longlong_as_double(((long long)lhi << 32) | (unsigned)llo) != sw) | 1ull;     // run starts
            }
            // ---- the next position's per-lane part (its header has arrived behind the network) ----
            if (ZT_U(Hn.seq) == t + 2) {
                const unsigned mn = (unsigned)ZT_U(Hn.meta);
                if ((mn >> 24) & 1u) sp_lanes_load(lds[(mn >> 20) & 0xfu], idx, Hn);
            }
            bool bad = in1 && ((ckey == nkey && cts == tnow) || cw != cw);
            if (__builtin_expect(__ballot(cand >= 0) != 0ull, 0)) {   // an occupied slot: compare the keys in full (partner entry j lives in lane 32 + j)
                const int src = 32 + (cand >= 0 ? cand : 0);
                const u64 kj = __shfl(ckey, src);
                const double tj = __shfl(cts, src);
                bad = bad || (in1 && cand >= 0 && kj == ckey && tj == cts);
            }
            fail = fail || __ballot(bad) != 0ull;
            // ---- the cut, on the scalar unit (all from the mask of run starts) ----
            const bool full = (S >> drop) & 1ull;               // the cut falls on a run start: exactly k candidates are kept
            const unsigned claimed = (unsigned)(S >> drop) & kmask;
            const u64 lowdrop = ((u64)2 << drop) - 1ull;        // positions 0 .. drop
            const int rsG = 63 - __builtin_clzll(S & lowdrop);  // start of the run that holds position `drop`
            const u64 nmask = ((u64)2 << ((n - 1) & 63)) - 1ull;   // positions 0 .. n-1 (n <= 63)
            const u64 multi = (~S | ~(S >> 1)) & nmask;         // position p shares its run with p-1 or with p+1
            const unsigned uo = (unsigned)(multi >> drop) & kmask;
            const u64 above = S & ~lowdrop;                     // the next run starts here (the padding's at n, at the latest)
            const int mo_ = full ? 0 : __ffsll((long long)above) - 1 - drop;
            const int na_ = full ? 0 : drop - rsG;
            const bool fin = full && claimed == kmask;          // all kept weights distinct: nothing provisional
            if (__builtin_expect(munc > 0, 0)) {
                // a picked member of the previous hop's straddling run that is kept here (or ties with the cut) needs the
                // previous hop's replay first: the helper's general code waits for it
                const int thr = full ? drop : rsG;
                const int lt = 63 - __builtin_clzll(S & (((u64)2 << sp) - 1ull));
                fail = fail || __ballot(lane < munc && lt >= thr) != 0ull;
            }
            const int ps = sp - drop;
            const bool kept = (unsigned)ps < (unsigned)k;       // (padding lanes sort behind position n-1)
            if (__builtin_expect(!fail && t >= MAIL_R && ZT_U(fs) != t - MAIL_R + 1, 0)) {
                __builtin_amdgcn_s_setprio(1);
                if (!wait_seq(&out_slot->seq_free, t - MAIL_R + 1, status, -5, t)) return false;
                __builtin_amdgcn_s_setprio(3);
            }
            go = !fail;
            if (__builtin_expect(go, 1)) {
                // ---- keys and time stamps of the new row: the candidate kept at provisional slot ps moves to lane ps ----
                const int dest = kept ? ps : 63;
                if (DUO) hw = __longlong_as_double((long long)push_u64((u64)__double_as_longlong(cw), dest));   // (no network here: the weights move like the keys)
                hkey = push_u64(ckey, dest);
                hts = __longlong_as_double((long long)push_u64((u64)__double_as_longlong(cts), dest));
                // ---- the mailbox slot, as a lean hop writes it: the helpers' off-chain halves and the fallback read it ----
                if (kept) { out_slot->key[ps] = ckey; out_slot->ts[ps] = cts; out_slot->w[ps] = cw; }
                if (!full && sp >= rsG && sp < drop) out_slot->alt_key[sp - rsG] = ckey;
                if (fin && kept) out_slot->pos[ps] = ps;
                // ---- ... and what this position's helper needs for the rest of the hop ----
                if (!DUO) Lw.sort.r[lane] = sp;
                if (lane == 0) {
                    mail_v4u v;
                    v.x = Hc.b.x; v.y = Hc.b.y;
                    v.z = (unsigned)k | ((unsigned)mo_ << 8) | ((unsigned)na_ << 16) | (1u << 24);
                    v.w = uo;
                    *reinterpret_cast<mail_v4u *>(&out_slot->norm) = v;
                    if (!DUO) P->S = S;
                    asm volatile("" ::: "memory");
                    __hip_atomic_store(&out_slot->seq_set, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (fin) __hip_atomic_store(&out_slot->seq_ord, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_store(&P->res, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                h0 = (unsigned)ZT_U(Hc.b.x); h1 = (unsigned)ZT_U(Hc.b.y);
                n1 = k; munc = mo_; nalt = na_; sorted = 1;
                ++n_ran;
            }
        }
        if (__builtin_expect(!go, 0)) {
            if (lane == 0) __hip_atomic_store(&P->res, -(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            have = false;
            if (DUO) {
                // the weights wave's records from here on are void: it picks the row up where this wave will
                ++duo_gen;
                if (lane == 0) {
                    __hip_atomic_store(&mail->a_restart, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    asm volatile("" ::: "memory");
                    __hip_atomic_store(&mail->a_gen, duo_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        return true;
    };
    __builtin_amdgcn_s_setprio(3);
    for (int t = 0; t < len; t += 2) {
        if (!hop(t, HA, HB)) break;
        if (t + 1 < len && !hop(t + 1, HB, HA)) break;
    }
    if (DUO && lane == 0) __hip_atomic_store(&mail->a_restart, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // the weights wave may go
    chain_stat(h.ctl, lane, ST_PAIR_DONE, n_ran);
    chain_stat(h.ctl, lane, ST_PAIR_BAIL_CRIT, len - n_ran);
#undef ZT_U
}

// Duo mode: the WEIGHTS wave.  What makes hop t + 1 wait for hop t is the hub row's weights alone -- scaled, merged with the
// partner's sorted side by the network, the kept ones shifted down -- ~175 of the hop's ~440 instructions; keys, time stamps, the
// tests and the publication can trail.  This wave runs that recurrence and nothing else, AHEAD of the spine: it assumes every
// test of the lean section will pass, leaves the sorted positions (in the helper's WaveLds) and the mask of run starts (in the
// position's prep record) under its generation number, and moves on.  The spine (chain_spine<true>) takes them instead of
// running the network itself; whenever it leaves a position to its helper it bumps the generation and names the position to
// restart from -- this wave drops what it has, waits for that position's slot like the spine does, and picks the weights up
// there.  A position whose side is not prepared, or whose shape the recurrence does not cover, makes it wait for exactly that.
__device__ __attribute__((always_inline)) inline void chain_weights(const zt_tppr &h, WaveLds *lds, int lane, Mail *mail, int len)
{
    const int k = h.k;
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    const bool low = __builtin_amdgcn_inverse_ballot_w64(0xffffffffull);
    const int idx = lane & 31;
    int *status = h.ctl + 2;
#define ZT_U(x) __builtin_amdgcn_readfirstlane((int)(x))
    double hw = 0.0;
    int n1 = 0, gen = 0, t = 0;
    bool have = false;
    __builtin_amdgcn_s_setprio(3);
    // wait until the spine says where to (re)start: a_gen != gen
    auto wait_restart = [&]() -> bool {
        unsigned spins = 0;
        for (;;) {
            const int g = lds_load_seq(&mail->a_gen);
            if (g != gen) {
                asm volatile("" ::: "memory");
                gen = g;
                t = ZT_U(lds_load_seq(&mail->a_restart));
                have = false;
                return t >= 0;                                    // (< 0: the spine has finished meanwhile)
            }
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 4095u) == 0 && launch_failed(status)) return false;
            // the chain is over once its head has passed the end and the spine is gone: the spine always bumps the generation
            // for position 0, so a chain of length >= 1 gets here at least once; a finished chain: see below
            if (lds_load_seq(&mail->a_restart) < 0) return false;
        }
    };
    if (!wait_restart()) return;                                  // (position 0 is always the helper's)
    for (;;) {
        // past the chain's end: the spine may still send this wave back, or finish (a_restart < 0)
        if (t >= len || ZT_U(lds_load_seq(&mail->a_gen)) != gen) { if (!wait_restart()) return; continue; }
        PrepHdr *P = &mail->prep[t % PREP_R];
        if (!have) {
            MailSlot *in_slot = &mail->slot[(t + MAIL_R - 1) % MAIL_R];
            // the row as position t - 1 was published (by its helper: that is why we are here)
            unsigned spins = 0;
            bool again = false;
            while (lds_load_seq(&in_slot->seq_set) != t) {
                __builtin_amdgcn_s_sleep(1);
                if (lds_load_seq(&mail->a_gen) != gen) { again = true; break; }
                if ((++spins & 4095u) == 0 && launch_failed(status)) return;
            }
            if (again) continue;
            asm volatile("" ::: "memory");
            double hn; int a0, a1, a2, a3; unsigned a4;
            mail_hdr_read(in_slot, hn, a0, a4, a1, a2, a3);
            hw = in_slot->w[idx];
            n1 = ZT_U(a0);
            have = true;
        }
        {   // the prepared side of position t
            unsigned spins = 0;
            bool again = false;
            while (lds_load_seq(&P->seq) != t + 1) {
                __builtin_amdgcn_s_sleep(1);
                if (lds_load_seq(&mail->a_gen) != gen) { again = true; break; }
                if ((++spins & 4095u) == 0 && launch_failed(status)) return;
            }
            if (again) continue;
            asm volatile("" ::: "memory");
        }
        const unsigned meta = (unsigned)ZT_U(P->meta);
        const int nb = (int)(meta & 0xffu);
        const int n = n1 + nb, drop = n - k;
        if (((meta >> 24) & 1u) == 0u || n1 <= 0 || drop <= 0 || n > 63) {
            // not a position of the lean kind: the spine will leave it to its helper and say where to go on
            if (!wait_restart()) return;
            continue;
        }
        WaveLds &Lw = lds[(meta >> 20) & 0xfu];
        const double scale_s1 = P->scale_s1;
        double sw = __longlong_as_double((long long)Lw.key[64 + idx]);
        int sid = Lw.sel[idx];
        const bool in1 = lane < n1;
        if (low) { const double cw = hw * scale_s1; sw = in1 ? cw : inf; sid = lane; }
        merge_stage<32>(sw, sid);
        merge_stage<16>(sw, sid);
        merge_stage<8>(sw, sid);
        merge_stage<4>(sw, sid);
        merge_stage<2>(sw, sid);
        merge_stage<1>(sw, sid);
        const long long swb = __double_as_longlong(sw);
        {   // the next row's weights: sorted lane j + drop -> lane j
            const int from = (lane + drop) << 2;
            const unsigned wl = (unsigned)__builtin_amdgcn_ds_bpermute(from, (int)(unsigned)(swb & 0xffffffffll));
            const unsigned wh = (unsigned)__builtin_amdgcn_ds_bpermute(from, (int)(swb >> 32));
            hw = __longlong_as_double((long long)(((u64)wh << 32) | wl));
        }
        const int sp = push_i32(lane, sid);
        const int llo = __builtin_amdgcn_mov_dpp((int)(unsigned)(swb & 0xffffffffll), 0x138, 0xf, 0xf, true);   // wave_shr:1
        const int lhi = __builtin_amdgcn_mov_dpp((int)(swb >> 32), 0x138, 0xf, 0xf, true);
        const u64 S = __ballot(__longlong_as_double(((long long)lhi << 32) | (unsigned)llo) != sw) | 1ull;
        Lw.sort.r[lane] = sp;
        if (lane == 0) {
            P->S = S;
            asm volatile("" ::: "memory");
            __hip_atomic_store(&P->a_seq, (gen << 12) | (t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        n1 = k;
        ++t;
    }
#undef ZT_U
}

