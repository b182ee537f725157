// The spine's loop (chain_spine, zebra_amd/csrc/tppr_chain.hpp) ALONE: one workgroup, wave 0 = the spine, wave 1 = a
// feeder (wave 2) that plays all the helpers (posts a prepared side per position ahead of the spine, frees ring slots, runs the
// hops the spine leaves to "the helper" by publishing a fresh row), waves 2..7 idle or -- noise = 1 -- busy with LDS and
// vector work.  Prints core clocks per hop: what the spine's instruction stream costs with nothing in its way.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o tools/out/spine_bench tools/exp/spine_bench.hip && tools/out/spine_bench [hops] [noise]
#include "../../zebra_amd/csrc/tppr_chain.hpp"

namespace {
thread_local char g_err[256];
}
void zt::set_error(const char *, ...) {}

__global__ __launch_bounds__(512) void k_bench(zt_tppr h, int len, long long *out, int noise)
{
    __shared__ WaveLds lds[WAVES_PER_WG];
    __shared__ Mail mail;
    __shared__ int done;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < MAIL_R) { mail.slot[threadIdx.x].seq_set = 0; mail.slot[threadIdx.x].seq_ord = 0; mail.slot[threadIdx.x].seq_free = 0; }
    if (threadIdx.x < PREP_R) { mail.prep[threadIdx.x].seq = 0; mail.prep[threadIdx.x].res = 0; }
    if (threadIdx.x == 0) { mail.head = 0; done = 0; mail.a_gen = 0; mail.a_restart = 0; }
    if (threadIdx.x < PREP_R) mail.prep[threadIdx.x].a_seq = 0;
    for (int q = threadIdx.x; q < WAVES_PER_WG * HTAB; q += blockDim.x) lds[q / HTAB].htab[q % HTAB] = -1;
    __syncthreads();
    const int k = h.k;
    const double beta = h.beta[0];
    if (wave == 0) {
        const long long t0 = (long long)__builtin_readcyclecounter();
#ifdef SB_DUO
        chain_spine<true>(h, lds, lane, &mail, len);
#else
        chain_spine<false>(h, lds, lane, &mail, len);
#endif
        const long long t1 = (long long)__builtin_readcyclecounter();
        if (lane == 0) { out[0] = t1 - t0; __hip_atomic_store(&done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#ifdef SB_DUO
    } else if (wave == 1) {
        chain_weights(h, lds, lane, &mail, len);
#endif
    } else if (wave == 2) {
        WaveLds &L = lds[2];
        const int nb = k + 1;
        const double inf = __longlong_as_double(0x7ff0000000000000ll);
        // the partner's side, the same at every position: nb candidates with weights 0.011 .. descending in sorted lanes
        if (lane >= 32) {
            const int j = lane - 32;                                  // home lane 32 + j: candidate j
            L.key[j] = 0x100000000ull * (1000 + j) + 77; L.ts[j] = 5.0 + j; L.w[j] = j < nb ? 0.011 + 0.0007 * ((j * 7) % nb) : 0.0;
            // sorted descending behind +inf padding: sorted lane 64 - nb + r holds the r-th largest
            const int r = lane - (64 - nb);
            double swv = inf; int sidv = lane + nb;                   // padding: an unused home lane
            if (r >= 0) {
                // r-th largest of 0.011 + 0.0007 * perm(j): perm(j) = (7 j) mod nb is a permutation (nb = 21: gcd(7, 21) != 1 -> use j itself then)
                swv = 0.0; sidv = 32;
            }
            L.key[64 + j] = (u64)__double_as_longlong(swv); L.sel[j] = sidv;
        }
        wave_sync();
        // (weights by rank, done by lane 0 for clarity)
        if (lane == 0) {
            double w[32]; int id[32];
            for (int j = 0; j < nb; ++j) { w[j] = 0.011 + 0.0007 * j + 0.00001 * ((j * 5) % 7); id[j] = j; L.w[j] = w[j]; }
            for (int a = 0; a < nb; ++a) for (int b = a + 1; b < nb; ++b) if (w[b] > w[a]) { double x = w[a]; w[a] = w[b]; w[b] = x; int y = id[a]; id[a] = id[b]; id[b] = y; }
            for (int r = 0; r < nb; ++r) { const int sl = 64 - nb + r - 32; L.key[64 + sl] = (u64)__double_as_longlong(w[r]); L.sel[sl] = 32 + id[r]; }
        }
        wave_sync();
        double pn = 3.0;                                              // the hub's norm at position 0
        auto publish_row = [&](int t, double norm_after) {
            MailSlot *sl = &mail.slot[t % MAIL_R];
            if (lane < k) { sl->key[lane] = 0x100000000ull * (5000 + t * 32 + lane) + 9; sl->ts[lane] = 1.0 + lane; sl->w[lane] = 0.01 + 0.001 * lane; }
            if (lane == 0) mail_hdr_write(sl, norm_after, k, 0u, 0, 0, 1);
            if (lane < k) sl->pos[lane] = lane;
            asm volatile("" ::: "memory");
            if (lane == 0) { __hip_atomic_store(&sl->seq_set, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                             __hip_atomic_store(&sl->seq_ord, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
        };
        int posted = 0, seen = 0, left = 0;
        double pn_post = pn;                                          // norm arriving at position `posted`
        while (seen < len) {
            while (posted < len && posted < seen + PREP_R - 1) {
                PrepHdr *P = &mail.prep[posted % PREP_R];
                const double nn = pn_post * beta + beta;
                if (lane == 0) {
                    __hip_atomic_store(&P->res, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    P->norm = pn_post; P->scale_s1 = 0.93; P->norm_next = nn; P->tnow = 100.0 + posted;      // (no division: the feeder must be faster than the spine)
                    P->nkey = 0x100000000ull * (900000 + posted) + 3;
                    P->meta = posted == 0 ? 0u : ((unsigned)nb | ((unsigned)k << 8) | (1u << 16) | (2u << 20) | (1u << 24));
                    asm volatile("" ::: "memory");
                    __hip_atomic_store(&P->seq, posted + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                pn_post = nn;
                ++posted;
            }
            int r;
            {
                const int *rp = &mail.prep[seen % PREP_R].res;
                unsigned spins = 0;
                for (;;) {
                    r = lds_load_seq(rp);
                    if (r == seen + 1 || r == -(seen + 1)) break;
#ifdef ZT_FEED_SLEEP
                    __builtin_amdgcn_s_sleep(ZT_FEED_SLEEP);
#endif
                    if ((++spins & 0xfffffu) == 0) { r = 0; break; }
                }
            }
            if (r == 0) break;
            pn = pn * beta + beta;                                    // the norm after position `seen`
            if (r < 0) { publish_row(seen, pn); ++left; }
            if (seen >= 1 && lane == 0) __hip_atomic_store(&mail.slot[(seen - 1) % MAIL_R].seq_free, seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            ++seen;
        }
        if (lane == 0) out[1] = left;
    } else if (noise && wave >= 3) {
        WaveLds &L = lds[wave];
        double acc = lane;
        while (__hip_atomic_load(&done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) {
            for (int q = 0; q < 64; ++q) { L.sort.v[(lane + q) & 127] = acc; acc = acc * 1.0000001 + L.sort.v[(lane * 3 + q) & 127]; }
        }
        if (acc == 12345.678) out[2] = 1;
    }
}

int main(int argc, char **argv)
{
    const int len = argc > 1 ? atoi(argv[1]) : 1500, noise = argc > 2 ? atoi(argv[2]) : 0;
    zt_tppr h;
    memset(&h, 0, sizeof(h));
    h.k = 20; h.M = 1; h.alpha[0] = 0.1; h.beta[0] = 0.5;
    hipMalloc(&h.ctl, sizeof(int) * CTL_WORDS);
    hipMemset(h.ctl, 0, sizeof(int) * CTL_WORDS);
    long long *out, ho[4] = {0, 0, 0, 0};
    hipMalloc(&out, sizeof(ho));
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(out, 0, sizeof(ho));
        k_bench<<<1, 512>>>(h, len, out, noise);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(ho, out, sizeof(ho), hipMemcpyDeviceToHost);
        int ctl[16];
        hipMemcpy(ctl, h.ctl, sizeof(ctl), hipMemcpyDeviceToHost);
        printf("%s: %d hops, %.0f clocks per hop (%lld left to the helper), sections run by the spine %d, status %d\n", hipGetErrorString(e), len, (double)ho[0] / len, ho[1],
               ctl[ST_PAIR_DONE], ctl[2]);
        hipMemset(h.ctl, 0, sizeof(int) * CTL_WORDS);
    }
    return 0;
}
