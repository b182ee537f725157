// TGN node-memory maintenance (P3): "last message wins" raw-message store and
// the GRU memory update.
//
// Replaces TGN.get_raw_messages + Memory.store_raw_messages
// (reference model/tgn_model.py:204-226, modules/memory.py:27-30) and
// SequenceMemoryUpdater.update_memory / update_memory_in_test with nn.GRUCell
// (modules/memory_updater.py:29-57,95-98) + Memory.clear_messages (:59-60).
//
// The reference resolves "last occurrence per node" on the host (np.unique on
// the flipped batch) and keeps the pending-message flags in a host numpy
// array; here both stay on the device: an atomicMax over batch positions picks
// the winner, flags are a device u8 array.  The GRU runs on exact-f32 MFMA
// with the gathered [message | memory] rows staged in LDS.
#include "common.hpp"
#include "embed_out_body.hpp"     // the output layers of the embedding as a device function (k_out_gru below)

using namespace zt;

namespace {

// (f32x4, AGG_THREADS and the output layers' body: embed_out_body.hpp, included above)

__host__ __device__ inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// ------------------------------------------------------------ last message ----
__global__ void k_last_pos(const int *__restrict__ src, const int *__restrict__ dst,
                           const long long *__restrict__ eidx, long long B, long long num_nodes, long long num_edges,
                           int *scratch, int *status, int *zero_word)
{
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p == 0 && zero_word != nullptr) *zero_word = 0;      // (the row counter of the GRU update that follows on this stream)
    if (p >= 2 * B) return;
    const int v = p < B ? src[p] : dst[p - B];
    const long long e = eidx[p < B ? p : p - B];
    if (v < 0 || v >= num_nodes || e < 0 || e >= num_edges) { atomicExch(status, ZT_ERR_RANGE); return; }
    atomicMax(&scratch[v], (int)p);
}

// one wavefront per batch position; only the last occurrence of a node writes
__global__ __launch_bounds__(256) void k_build_messages(
    const float *__restrict__ memory, const float *__restrict__ last_update, const float *__restrict__ efeat,
    const float *__restrict__ time_w, long long num_nodes, long long num_edges, int D, int F, int T,
    const int *__restrict__ src, const int *__restrict__ dst, const double *__restrict__ ts,
    const long long *__restrict__ eidx, long long B, float *messages, float *msg_ts, unsigned char *flags,
    int *scratch, int *uniq_ids, int *n_uniq, const int *status, long long pos_lo, long long pos_hi, int set_flags)
{
    const long long p = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (p >= 2 * B) return;
    const bool rejected = *status == ZT_ERR_RANGE;          // whole call rejected: only restore scratch
    const long long i = p < B ? p : p - B;
    const int v = p < B ? src[i] : dst[i];
    if (v < 0 || v >= num_nodes) return;
    if (scratch[v] != (int)p) return;                        // not the last occurrence (or already reset)
    if (!rejected && p >= pos_lo && p < pos_hi) {
        const int partner = p < B ? dst[i] : src[i];
        const float tf = (float)ts[i];                       // edge_times .float() (tgn_model.py:213)
        const float delta = tf - last_update[v];             // :221
        const int msg = 2 * D + F + T;
        float *row = messages + (size_t)v * msg;
        const float *m1 = memory + (size_t)v * D, *m2 = memory + (size_t)partner * D;
        const float *er = efeat + (size_t)eidx[i] * F;
        for (int c = lane; c < D; c += WAVE) row[c] = m1[c];
        for (int c = lane; c < D; c += WAVE) row[D + c] = m2[c];
        for (int c = lane; c < F; c += WAVE) row[2 * D + c] = er[c];
        for (int c = lane; c < T; c += WAVE) row[2 * D + F + c] = time_cosf(delta * time_w[c]);
        if (lane == 0) {
            msg_ts[v] = tf;
            if (set_flags) flags[v] = 1;           // (0: the caller consumes the list of winners itself, see store_messages_ex)
            if (uniq_ids) uniq_ids[atomicAdd(n_uniq, 1)] = v;
            else if (n_uniq) atomicAdd(n_uniq, 1);
        }
    }
    if (lane == 0) scratch[v] = -1;
}

// The same, TWO batch positions per wavefront with every load of both in flight before the first store (D, T <= 128,
// F <= 256).  On the pipeline's message stream this kernel runs beside k_stream's workgroups, which leave room for ONE
// more wave per SIMD: the kernel's time is (positions) x (a wave's chain of dependent round trips) / (4 waves per CU), and
// a wave that carries two positions through that chain halves it (C5 on 64 CUs: 190 -> ~100 us).
__global__ __launch_bounds__(256) void k_build_messages2(
    const float *__restrict__ memory, const float *__restrict__ last_update, const float *__restrict__ efeat,
    const float *__restrict__ time_w, long long num_nodes, long long num_edges, int D, int F, int T,
    const int *__restrict__ src, const int *__restrict__ dst, const double *__restrict__ ts,
    const long long *__restrict__ eidx, long long B, float *messages, float *msg_ts, unsigned char *flags,
    int *scratch, int *uniq_ids, int *n_uniq, const int *status, long long pos_lo, long long pos_hi, int set_flags)
{
    const int lane = threadIdx.x & 63;
    const long long wv = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long half = (2 * B + 1) / 2;                  // wave w takes positions w and w + half
    if (wv >= half) return;
    const bool rejected = *status == ZT_ERR_RANGE;          // whole call rejected: only restore scratch
    const int msg = 2 * D + F + T;
    long long p[2] = {wv, wv + half};
    int v[2], partner[2];
    bool own[2], write[2];
    float tf[2], lu[2];
    long long e[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        own[q] = false; write[q] = false; v[q] = 0; partner[q] = 0; tf[q] = 0.f; lu[q] = 0.f; e[q] = 0;
        if (p[q] < 2 * B) {
            const long long i = p[q] < B ? p[q] : p[q] - B;
            const int a = src[i], b = dst[i];
            v[q] = p[q] < B ? a : b; partner[q] = p[q] < B ? b : a;
            tf[q] = (float)ts[i];                            // edge_times .float() (tgn_model.py:213)
            e[q] = eidx[i];
        }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
        if (p[q] < 2 * B && v[q] >= 0 && v[q] < num_nodes) { own[q] = scratch[v[q]] == (int)p[q]; lu[q] = last_update[v[q]]; }
#pragma unroll
    for (int q = 0; q < 2; ++q) write[q] = own[q] && !rejected && p[q] >= pos_lo && p[q] < pos_hi;
    // ---- all loads of both rows ----
    float a1[2][2], a2[2][2], ef[2][4], tw[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) tw[u] = (lane + 64 * u) < T ? time_w[lane + 64 * u] : 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const float *m1 = memory + (size_t)v[q] * D, *m2 = memory + (size_t)partner[q] * D;
        const float *er = efeat + (size_t)e[q] * F;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int c = lane + 64 * u;
            a1[q][u] = (write[q] && c < D) ? m1[c] : 0.f;
            a2[q][u] = (write[q] && c < D) ? m2[c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = lane + 64 * u;
            ef[q][u] = (write[q] && c < F) ? er[c] : 0.f;
        }
    }
    // ---- stores ----
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        if (write[q]) {
            float *row = messages + (size_t)v[q] * msg;
            const float delta = tf[q] - lu[q];               // :221
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int c = lane + 64 * u;
                if (c < D) { row[c] = a1[q][u]; row[D + c] = a2[q][u]; }
                if (c < T) row[2 * D + F + c] = time_cosf(delta * tw[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = lane + 64 * u;
                if (c < F) row[2 * D + c] = ef[q][u];
            }
            if (lane == 0) {
                msg_ts[v[q]] = tf[q];
                if (set_flags) flags[v[q]] = 1;
                if (uniq_ids) uniq_ids[atomicAdd(n_uniq, 1)] = v[q];
                else if (n_uniq) atomicAdd(n_uniq, 1);
            }
        }
        if (own[q] && lane == 0) scratch[v[q]] = -1;
    }
}

// ------------------------------------------------------------------- GRU ----
// Compact the flagged subset of ids (or of all nodes) into rows[]; clear the
// flags of every id considered (Memory.clear_messages).
__global__ void k_select_flagged(const int *__restrict__ ids, long long n_ids, const int *__restrict__ n_ids_dev,
                                 long long num_nodes, unsigned char *flags, int *rows, int *n_rows)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long n = ids ? (n_ids_dev ? (long long)*n_ids_dev : n_ids) : num_nodes;
    if (ids && n > n_ids) n = n_ids;
    bool take = false;
    int v = -1;
    if (t < n) {
        v = ids ? ids[t] : (int)t;
        if (v >= 0 && v < num_nodes) {
            // clear the flag byte atomically (32-bit word): a duplicated id is selected once
            unsigned *wp = reinterpret_cast<unsigned *>(flags + (v & ~3));
            const unsigned mask = 0xffu << (8 * (v & 3));
            take = (atomicAnd(wp, ~mask) & mask) != 0;
        }
    }
    // one counter add per wavefront (thousands of adds to one word cost ~12 ns each)
    const u64 bm = __ballot(take);
    if (bm == 0ull) return;
    int base = 0;
    const int lane = threadIdx.x & 63;
    if (lane == __ffsll((long long)bm) - 1) base = atomicAdd(n_rows, __popcll(bm));
    base = __shfl(base, __ffsll((long long)bm) - 1);
    if (take) rows[base + __popcll(bm & ((1ull << lane) - 1ull))] = v;
}

// k_gru<MT>, MT = 1:            16 rows per workgroup: the kernel's time is one workgroup's latency (<= 1 per CU),
                               // so smaller tiles on more CUs win over weight-fragment reuse -- also at C5's 8 192 rows
                               // (512 tiles, each streaming the 686 KB of gate weights from L2: k_gru<2>, measured slower)
constexpr int GRU_NTW = 1;     // hidden N-tiles per wave; 8 waves -> D <= 128
constexpr int GRU_WAVES = 8;
constexpr int GRU_CH = 6;      // k-steps of weight fragments in flight
constexpr int GRU_SRC_WORD = 32; // words 32, 33 of the workspace's counter block: the gate of k_out_gru / k_out_gru2 (SrcGate)

// The gate between the two halves of k_out_gru / k_out_gru2.  The output layers' source path reads memory[nodes] as the
// previous update left it and the GRU half rewrites rows of the same nodes: a source-path unit (a workgroup of k_out_gru, a
// wave of k_out_gru2) adds 1 to word[0] once its rows have arrived, and a GRU workgroup waits for word[0] == target before its
// first write to the table.  ALL of the protocol's state is these two words of the GRU workspace (round-5 advisor: a parity
// kept on the host could disagree with them -- a second pipeline or a direct zt_gru_update on the same workspace, a workspace
// re-packed in between): every participant -- source-path unit or GRU workgroup -- adds 1 to word[1] when it is through, and
// the last one out zeroes both words, so a launch always finds them at zero.  Launches that share a workspace must be
// ordered by a stream, as for the row list beside these words.  The wait is bounded like every wait of k_stream
// (tppr_rows.hpp): after GATE_TICKS it writes ZT_ERR_TIMEOUT to the caller's status word and, where the pipeline gave one,
// to a host-mapped latch that fails the next step call -- and the workgroup leaves the table untouched.
constexpr long long GATE_TICKS = 400000000ll;     // 4 s of the 100 MHz wall clock
struct SrcGate {
    int *word;
    unsigned target, participants;
    int *status, *latch;
};

// by ONE thread of a GRU workgroup; false: gave up (reported)
__device__ __forceinline__ bool gate_wait(const SrcGate &g)
{
    unsigned spins = 0;
    long long t0 = 0;
    while (ld_agent(g.word) < (int)g.target) {          // (signed: a word that is not a count of this launch's units never opens the gate)
        __builtin_amdgcn_s_sleep(2);
        if ((++spins & 2047u) == 0) {
            const long long now = (long long)wall_clock64();
            if (t0 == 0) t0 = now;
            else if (now - t0 > GATE_TICKS) {
                atomicExch(g.status, ZT_ERR_TIMEOUT);
                if (g.latch != nullptr) __hip_atomic_store(g.latch, (int)ZT_ERR_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                return false;
            }
        }
    }
    return true;
}

// by the thread that added to / waited for word[0], once per participant, when its unit is through
__device__ __forceinline__ void gate_leave(const SrcGate &g)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (this unit's add to word[0] has reached L2 before its add to word[1] can)
    const unsigned before = (unsigned)atomicAdd(g.word + 1, 1);
    if (before + 1u == g.participants) { st_agent(g.word, 0); st_agent(g.word + 1, 0); }
}

// Zero-padded gate-major copy in FRAGMENT order: W[3D][K] -> Wp[3][Dp / 16][Kp / 16][64 lanes][4]: the 16 x 16 block (N-tile
// nt, k-chunk kc) of a gate as the MFMA's lanes hold it -- lane (r16, g4) has W[16 nt + r16][16 kc + 4 g4 .. + 3] -- so that a
// wave's fragment load is ONE contiguous kilobyte.  (Row-major, the same load touched sixteen 128-byte lines and used half
// of each: k_gru streams all 559 KB of gate weights per 16-row tile from L2, 286 MB per launch at C5's batch.)
__global__ void k_pack_gates(const float *__restrict__ W, int D, int K, float *__restrict__ Wp, int Dp, int Kp)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * Dp * Kp) return;
    const int e = i & 3, lane = (i >> 2) & 63, rest = i >> 8;
    const int KC = Kp / 16, NT = Dp / 16;
    const int kc = rest % KC, nt = (rest / KC) % NT, g = rest / (KC * NT);
    const int r = 16 * nt + (lane & 15), c = 16 * kc + 4 * (lane >> 4) + e;
    Wp[i] = (r < D && c < K) ? W[((size_t)g * D + r) * K + c] : 0.f;
}

struct GruArgs {
    float *memory, *last_update;
    const float *messages, *msg_ts;
    const int *rows, *n_rows;
    int D, msg_dim, Xp, Hp, lda;
    const float *Wih_p, *Whh_p, *b_ih, *b_hh, *Wm_p;
    float *P;
    int cap;
};

// bid = the workgroup's 16 MT-row tile.  gate != nullptr (k_out_gru): before a row of the memory table is written every
// source-path workgroup of the output layers must have its rows in LDS (SrcGate; LDS: one word more behind the node ids).
template <int MT>
__device__ __forceinline__ void gru_body(const GruArgs &G, char *smem, int bid, const SrcGate *gate)
{
    float *memory = G.memory, *last_update = G.last_update;
    const float *__restrict__ messages = G.messages, *__restrict__ msg_ts = G.msg_ts;
    const int *__restrict__ rows = G.rows, *__restrict__ n_rows = G.n_rows;
    const int D = G.D, msg_dim = G.msg_dim, Xp = G.Xp, Hp = G.Hp, lda = G.lda, cap = G.cap;
    const float *__restrict__ Wih_p = G.Wih_p, *__restrict__ Whh_p = G.Whh_p, *__restrict__ b_ih = G.b_ih, *__restrict__ b_hh = G.b_hh;
    const float *__restrict__ Wm_p = G.Wm_p;
    float *__restrict__ P = G.P;
    float *A = reinterpret_cast<float *>(smem);      // [32][lda]: [message (Xp) | memory (Hp)]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = bid * (MT * 16);
    // the tile's node ids are requested TOGETHER with the row count, not after it (one dependent round trip less; an
    // entry beyond the count is a stale id that nobody dereferences: every gather below is masked by g < nr)
    const int id_spec = tid < MT * 16 ? __builtin_nontemporal_load(rows + (r0 + tid < cap ? r0 + tid : cap - 1)) : 0;
    const int total = *n_rows;
    if (r0 >= total) return;
    const int nr = (total - r0) < MT * 16 ? (total - r0) : MT * 16;
    const int Dp = Hp;

    // flat (row, column) gather, GU loads in flight per thread before any LDS store (see aggregate.hip)
    const int nthr = 64 * GRU_WAVES;
    constexpr int GU = 8;
    int *rid = reinterpret_cast<int *>(A + (size_t)MT * 16 * lda);     // this tile's node ids
    if (tid < MT * 16) rid[tid] = tid < nr ? id_spec : 0;
    __syncthreads();
    for (int f0 = tid; f0 < MT * 16 * Xp; f0 += nthr * GU) {
        float v[GU];
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int f = f0 + u * nthr;
            const int g = f / Xp, c = f - g * Xp;
            v[u] = (f < MT * 16 * Xp && g < nr && c < msg_dim) ? messages[(size_t)rid[g] * msg_dim + c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int f = f0 + u * nthr;
            const int g = f / Xp, c = f - g * Xp;
            if (f < MT * 16 * Xp) A[(size_t)g * lda + c] = v[u];
        }
    }
    {   // the memory columns: all of a thread's elements in flight before its first LDS store (one round trip, not four)
        constexpr int HU = 4;
        for (int f0 = tid; f0 < MT * 16 * Hp; f0 += nthr * HU) {
            float v[HU];
#pragma unroll
            for (int u = 0; u < HU; ++u) {
                const int f = f0 + u * nthr, g = f / Hp, c = f - g * Hp;
                v[u] = (f < MT * 16 * Hp && g < nr && c < D) ? memory[(size_t)rid[g] * D + c] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < HU; ++u) {
                const int f = f0 + u * nthr, g = f / Hp, c = f - g * Hp;
                if (f < MT * 16 * Hp) A[(size_t)g * lda + Xp + c] = v[u];
            }
        }
    }
    __syncthreads();

    const int NT = (D + 15) / 16;
    const int r16 = lane & 15, g4 = lane >> 4;
    // per (m-tile, n-tile): r, z (message + memory), n_i (message), n_h (memory)
    f32x4 ar[MT][GRU_NTW], az[MT][GRU_NTW], ani[MT][GRU_NTW], anh[MT][GRU_NTW];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < GRU_NTW; ++b) {
            ar[a][b] = f32x4{0.f, 0.f, 0.f, 0.f}; az[a][b] = ar[a][b]; ani[a][b] = ar[a][b]; anh[a][b] = ar[a][b];
        }
    bool live[GRU_NTW];
    int colrow[GRU_NTW];
#pragma unroll
    for (int b = 0; b < GRU_NTW; ++b) {
        const int nt = wave + b * GRU_WAVES;
        live[b] = nt < NT;
        colrow[b] = (live[b] ? nt : 0) * 16 + r16;
    }
    // Both parts stream their weight fragments from L2; GRU_CH k-steps are fetched together so that one
    // round trip is paid per chunk rather than per k-step (a workgroup owns one tile, nothing else hides it).
    auto part = [&](const float *__restrict__ Wp, int Kp, int a_off, bool hidden) {
        const int KC = Kp / 16;
        for (int kc0 = 0; kc0 < KC; kc0 += GRU_CH) {
            f32x4 wr[GRU_CH][GRU_NTW], wz[GRU_CH][GRU_NTW], wn[GRU_CH][GRU_NTW];
#pragma unroll
            for (int c = 0; c < GRU_CH; ++c)
#pragma unroll
                for (int b = 0; b < GRU_NTW; ++b) {
                    const bool on = live[b] && kc0 + c < KC;
                    const size_t o = (((size_t)(colrow[b] >> 4) * KC + (on ? kc0 + c : 0)) * 64 + lane) * 4;       // (fragment order: k_pack_gates)
                    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
                    wr[c][b] = on ? *reinterpret_cast<const f32x4 *>(Wp + o) : zero;
                    wz[c][b] = on ? *reinterpret_cast<const f32x4 *>(Wp + (size_t)Dp * Kp + o) : zero;
                    wn[c][b] = on ? *reinterpret_cast<const f32x4 *>(Wp + (size_t)2 * Dp * Kp + o) : zero;
                }
#pragma unroll
            for (int c = 0; c < GRU_CH; ++c) {
                if (kc0 + c >= KC) break;
                f32x4 av[MT];
#pragma unroll
                for (int a = 0; a < MT; ++a)
                    av[a] = *reinterpret_cast<const f32x4 *>(A + (size_t)(a * 16 + r16) * lda + a_off + 16 * (kc0 + c) + 4 * g4);
#pragma unroll
                for (int b = 0; b < GRU_NTW; ++b) {
                    if (!live[b]) continue;
#pragma unroll
                    for (int a = 0; a < MT; ++a)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            ar[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a][j], wr[c][b][j], ar[a][b], 0, 0, 0);
                            az[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a][j], wz[c][b][j], az[a][b], 0, 0, 0);
                            if (hidden) anh[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a][j], wn[c][b][j], anh[a][b], 0, 0, 0);
                            else        ani[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a][j], wn[c][b][j], ani[a][b], 0, 0, 0);
                        }
                }
            }
        }
    };
    part(Wih_p, Xp, 0, false);      // message part: gi = W_ih x
    part(Whh_p, Hp, Xp, true);      // memory part:  gh = W_hh h
    // (the new rows go back into the tile for the projection below: every wave must be done READING the old ones -- the
    //  memory columns are the K operand of everybody's W_hh product)
    if (gate != nullptr) {
        if (tid == 0) rid[MT * 16] = gate_wait(*gate) ? 1 : 0;
        __syncthreads();
        if (rid[MT * 16] == 0) return;                 // (gave up: reported; the table keeps its rows)
    } else if (P != nullptr) __syncthreads();
    // gates (torch.nn.GRUCell): r,z = sigmoid(gi+gh); n = tanh(gi_n + r*gh_n); h' = (1-z)*n + z*h
#pragma unroll
    for (int b = 0; b < GRU_NTW; ++b) {
        if (!live[b]) continue;
        const int col = (wave + b * GRU_WAVES) * 16 + r16;
        if (col >= D) continue;
        const float bir = b_ih[col], biz = b_ih[D + col], bin = b_ih[2 * D + col];
        const float bhr = b_hh[col], bhz = b_hh[D + col], bhn = b_hh[2 * D + col];
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int g = a * 16 + g4 * 4 + j;
                if (g >= nr) continue;
                const float r = 1.f / (1.f + expf(-(ar[a][b][j] + bir + bhr)));
                const float z = 1.f / (1.f + expf(-(az[a][b][j] + biz + bhz)));
                const float n = tanhf(ani[a][b][j] + bin + r * (anh[a][b][j] + bhn));
                const float hold = A[(size_t)g * lda + Xp + col];
                const float hnew = (1.f - z) * n + z * hold;
                memory[(size_t)rid[g] * D + col] = hnew;
                if (P != nullptr) A[(size_t)g * lda + Xp + col] = hnew;    // (this thread alone reads and writes the element)
            }
    }
    for (int g = tid; g < nr; g += nthr) {
        const int v = rid[g];
        last_update[v] = msg_ts[v];               // memory_updater.py:40
    }
    // ---- the projected table follows the rows just rewritten: P[v] = W_m memory'[v] (aggregate.hip, k_project_rows),
    // ---- here from the new rows while they are still in LDS: one kernel and one pass over the rows less per step
    if (P != nullptr) {
        __syncthreads();                              // every column of the new rows is in the tile
        if (wave < Hp / 16) {                         // wave w: output columns 16 w .. 16 w + 15
            const int KC = Hp / 16;                   // <= 8 (D <= 128)
            f32x4 wv[8];
#pragma unroll
            for (int c = 0; c < 8; ++c)
                wv[c] = c < KC ? *reinterpret_cast<const f32x4 *>(Wm_p + (size_t)(wave * 16 + r16) * Hp + 16 * c + 4 * g4)
                               : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    if (c >= KC) break;
                    const f32x4 av = *reinterpret_cast<const f32x4 *>(A + (size_t)(a * 16 + r16) * lda + Xp + 16 * c + 4 * g4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], wv[c][j], acc, 0, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int g = a * 16 + g4 * 4 + j;
                    if (g < nr) P[(size_t)rid[g] * Hp + wave * 16 + r16] = acc[j];
                }
            }
        }
    }
}

template <int MT>
__global__ __launch_bounds__(64 * GRU_WAVES) void k_gru(GruArgs G)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gru_body<MT>(G, smem, blockIdx.x, nullptr);
}

// The output layers and the GRU update in ONE launch (round 5).  The two kernels are independent but for the memory rows the
// output layers' source path reads, each is bound by one workgroup's chain of memory round trips and fills a fraction of the
// chip: one after the other they cost the step both latencies and a launch gap.  Workgroup order: [0, out_tiles) the source
// path of k_embed_out's body (they are dispatched first, so the GRU half's wait for their reads can never be a wait for a
// workgroup that has no compute unit), then the GRU tiles (the longest chains: not behind 200 short workgroups' dispatch),
// then the neighbour paths.
template <int HG>
__global__ __launch_bounds__(64 * GRU_WAVES) void k_out_gru(EmbedOutArgs E, int out_tiles, int gru_wgs, GruArgs G, SrcGate gate)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int bid = blockIdx.x;
    if (bid >= out_tiles && bid < out_tiles + gru_wgs) {
        gru_body<1>(G, smem, bid - out_tiles, &gate);
        if (threadIdx.x == 0) gate_leave(gate);              // (thread 0 is the one that waited, if the tile had rows at all)
        return;
    }
    if (threadIdx.x >= AGG_THREADS) return;                // (the body is written for four waves; a finished wave leaves the barriers)
    const int q = bid < out_tiles ? bid : bid - gru_wgs;     // tile + out_tiles * path
    embed_out_body<HG>(E, smem, q % out_tiles, q / out_tiles, gate.word);
    if (bid < out_tiles && threadIdx.x == 0) gate_leave(gate);   // (a source-path workgroup: thread 0 made its add to word[0])
}

// ---------------------------------------------------------------------------------------------------------
// k_gru_split (round 4): the same update for SMALL row counts.  Phase stamps of k_gru at C2's batch (400 rows, 25
// workgroups on 224 CUs; tools/exp/gru_phases.py): 66 000 cycles per workgroup, 49 000 of them in the two gate products
// -- a tile's 3 108 MFMAs all run on ONE compute unit (25 000 cycles of matrix pipe on its four SIMDs) while nine tenths
// of the chip idle.  Here a workgroup owns (16 rows, ONE N-tile of the hidden layer): seven times the workgroups, and
// its four waves split K, each with ALL its weight fragments (three gates x <= 10 chunks) requested before anything is
// looked at, beside the staging of the tile's [message | memory] rows (every element in flight before the first LDS
// store).  The waves' partial gate sums meet in LDS, wave 0 applies the gates.
// The N-tiles of a row are now computed by DIFFERENT workgroups, each of which reads the row's old memory columns as K
// input: nobody may write memory until all have read.  So the new values go to a staging buffer (workspace), and the
// LAST workgroup of a tile to arrive (a counter per tile, reset by that workgroup) commits the 16 rows to the memory
// table and projects them (P[v] = W_m h'[v] needs a row's all N-tiles anyway).
// ---------------------------------------------------------------------------------------------------------
#ifdef ZT_GRU_STAMP
// diagnostic build only (tools/exp/gru_phases.py): shader-clock readings at the phase boundaries of k_gru_split
__device__ unsigned long long g_gru[2048 * 10];
#define GSTAMP(i) do { if (tid == 0 && blockIdx.y == 0 && blockIdx.x < 2048) g_gru[blockIdx.x * 10 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define GSTAMP(i) do { } while (0)
#endif
constexpr int GS_WAVES = 4;
constexpr int GS_MAXCH = 10;         // chunks of 16 k-columns per wave: (Xp + Hp) / 16 <= 40 (msg <= 512)
constexpr int GS_STAGE = 37;         // staged elements per thread: 16 (msg + D) <= 37 * 256
constexpr int GS_TILE_COUNTERS = 512;
constexpr int GS_MAX_ROWS = GS_TILE_COUNTERS * 16;

struct GruSplitArgs {
    GruArgs g;
    int *tile_cnt;
    float *hnew;
};

// (bx, by) = (16-row tile, N-tile).  gate != nullptr (k_out_gru2): the tile's last workgroup waits at the gate (SrcGate)
// before it commits the rows to the memory table.
__device__ __forceinline__ void gru_split_body(const GruSplitArgs &GS, char *smem, int bx, int by, const SrcGate *gate)
{
    const GruArgs &G = GS.g;
    float *memory = G.memory, *last_update = G.last_update;
    const float *__restrict__ messages = G.messages, *__restrict__ msg_ts = G.msg_ts;
    const int *__restrict__ rows = G.rows, *__restrict__ n_rows = G.n_rows;
    const int D = G.D, msg_dim = G.msg_dim, Xp = G.Xp, Hp = G.Hp, lda = G.lda, cap = G.cap;
    const float *__restrict__ Wih_p = G.Wih_p, *__restrict__ Whh_p = G.Whh_p, *__restrict__ b_ih = G.b_ih, *__restrict__ b_hh = G.b_hh;
    const float *__restrict__ Wm_p = G.Wm_p;
    float *__restrict__ P = G.P;
    int *tile_cnt = GS.tile_cnt;
    float *hnew = GS.hnew;
    float *A = reinterpret_cast<float *>(smem);                          // [16][lda]: [message (Xp) | memory (Hp)], zero padded
    float *red = A + 16 * lda;                                           // [4 waves][4 sums][64 lanes][4]
    int *rid = reinterpret_cast<int *>(red + GS_WAVES * 4 * 64 * 4);     // [16] node ids; [16] = "this workgroup is the tile's last"
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g4 = lane >> 4;
    GSTAMP(0);
    const int r0 = bx * 16, nt = by, NT = Hp / 16;
    // (ids requested together with the row count: see k_gru)
    const int id_spec = tid < 16 ? __builtin_nontemporal_load(rows + (r0 + tid < cap ? r0 + tid : cap - 1)) : 0;
    const int total = *n_rows;
    if (r0 >= total) return;
    const int nr = (total - r0) < 16 ? (total - r0) : 16;
    GSTAMP(1);
    const int KCx = Xp / 16, KC = KCx + Hp / 16;
    const int col = 16 * nt + r16;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // ---- this wave's weight fragments: chunks wave, wave + 4, ... of [message | memory] for the three gates ----
    f32x4 wr[GS_MAXCH], wz[GS_MAXCH], wn[GS_MAXCH];
#pragma unroll
    for (int q = 0; q < GS_MAXCH; ++q) {
        const int c = wave + GS_WAVES * q;
        const bool on = c < KC, hid = c >= KCx;
        const float *W = hid ? Whh_p : Wih_p;
        const int Kp = hid ? Hp : Xp, cc = hid ? c - KCx : c;
        const size_t o = (((size_t)nt * (Kp / 16) + (on ? cc : 0)) * 64 + lane) * 4;                           // (fragment order: k_pack_gates)
        wr[q] = on ? *reinterpret_cast<const f32x4 *>(W + o) : zero4;
        wz[q] = on ? *reinterpret_cast<const f32x4 *>(W + (size_t)Hp * Kp + o) : zero4;
        wn[q] = on ? *reinterpret_cast<const f32x4 *>(W + (size_t)2 * Hp * Kp + o) : zero4;
    }
    const bool cin = col < D;
    const float bir = cin ? b_ih[col] : 0.f, biz = cin ? b_ih[D + col] : 0.f, bin = cin ? b_ih[2 * D + col] : 0.f;
    const float bhr = cin ? b_hh[col] : 0.f, bhz = cin ? b_hh[D + col] : 0.f, bhn = cin ? b_hh[2 * D + col] : 0.f;
    if (tid < 16) rid[tid] = tid < nr ? id_spec : 0;
    for (int f = tid; f < 16 * lda; f += 256) A[f] = 0.f;                // padding columns, rows beyond nr
    __syncthreads();
    GSTAMP(2);
    // ---- stage the tile: every load in flight before the first store ----
    const int W1 = msg_dim + D;
    {
        float v[GS_STAGE];
#pragma unroll
        for (int q = 0; q < GS_STAGE; ++q) {
            const int f = tid + q * 256, g = f / W1, c = f - g * W1;
            v[q] = g >= nr ? 0.f : (c < msg_dim ? messages[(size_t)rid[g] * msg_dim + c] : memory[(size_t)rid[g] * D + (c - msg_dim)]);
        }
#pragma unroll
        for (int q = 0; q < GS_STAGE; ++q) {
            const int f = tid + q * 256, g = f / W1, c = f - g * W1;
            if (g < nr) A[g * lda + (c < msg_dim ? c : Xp + (c - msg_dim))] = v[q];
        }
    }
    __syncthreads();
    GSTAMP(3);
    // ---- partial gate sums over this wave's chunks ([message | memory]: chunk c starts at column 16 c of the tile) ----
    f32x4 ar = zero4, az = zero4, ani = zero4, anh = zero4;
#pragma unroll
    for (int q = 0; q < GS_MAXCH; ++q) {
        const int c = wave + GS_WAVES * q;
        if (c >= KC) break;
        const bool hid = c >= KCx;
        const f32x4 av = *reinterpret_cast<const f32x4 *>(A + r16 * lda + 16 * c + 4 * g4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ar = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], wr[q][j], ar, 0, 0, 0);
            az = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], wz[q][j], az, 0, 0, 0);
            if (hid) anh = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], wn[q][j], anh, 0, 0, 0);
            else     ani = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], wn[q][j], ani, 0, 0, 0);
        }
    }
    GSTAMP(4);
    f32x4 *rw = reinterpret_cast<f32x4 *>(red) + (size_t)wave * 4 * 64;
    rw[0 * 64 + lane] = ar; rw[1 * 64 + lane] = az; rw[2 * 64 + lane] = ani; rw[3 * 64 + lane] = anh;
    __syncthreads();
    GSTAMP(5);
    if (wave == 0) {
        // gates (torch.nn.GRUCell): r, z = sigmoid(gi + gh); n = tanh(gi_n + r gh_n); h' = (1 - z) n + z h; the four waves'
        // partial sums are added in wave order
        const f32x4 *rr = reinterpret_cast<const f32x4 *>(red);
        f32x4 sr = rr[0 * 64 + lane], sz = rr[1 * 64 + lane], sni = rr[2 * 64 + lane], snh = rr[3 * 64 + lane];
#pragma unroll
        for (int wv = 1; wv < GS_WAVES; ++wv) {
            sr += rr[(wv * 4 + 0) * 64 + lane]; sz += rr[(wv * 4 + 1) * 64 + lane];
            sni += rr[(wv * 4 + 2) * 64 + lane]; snh += rr[(wv * 4 + 3) * 64 + lane];
        }
        if (cin) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int g = 4 * g4 + j;
                if (g >= nr) continue;
                const float r = 1.f / (1.f + expf(-(sr[j] + bir + bhr)));
                const float z = 1.f / (1.f + expf(-(sz[j] + biz + bhz)));
                const float n = tanhf(sni[j] + bin + r * (snh[j] + bhn));
                const float hold = A[g * lda + Xp + col];
                st_agent(reinterpret_cast<int *>(hnew + (size_t)(r0 + g) * Hp + col), __float_as_int((1.f - z) * n + z * hold));
            }
        }
        if (nt == 0 && lane < nr) { const int v = rid[lane]; last_update[v] = msg_ts[v]; }      // memory_updater.py:40
        // ---- arrive: the tile's last workgroup commits (sc1 stores drained, no fence: see k_gru_persist) ----
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int done = 0;
        if (lane == 0) done = atomicAdd(&tile_cnt[bx], 1);
        done = __builtin_amdgcn_readfirstlane(done);
        if (lane == 0) { rid[16] = done == NT - 1 ? 1 : 0; if (done == NT - 1) atomicExch(&tile_cnt[bx], 0); }
    }
    __syncthreads();
    GSTAMP(6);
    if (rid[16] == 0) return;
    if (gate != nullptr) {                                               // (k_out_gru2: the output layers' source path has read its rows)
        if (tid == 0) rid[17] = gate_wait(*gate) ? 1 : 0;
        __syncthreads();
        if (rid[17] == 0) return;                                        // (gave up: reported; the table keeps its rows)
    }
    // ---- every N-tile of these 16 rows has been computed from the OLD rows: new rows -> memory table, -> projected table
    // (read with sc1 loads: served by L2, where the other workgroups' drained sc1 stores are) ----
    {
        constexpr int CU = 7;                                            // 16 x Hp <= 7 * 256
        float v[CU];
#pragma unroll
        for (int q = 0; q < CU; ++q) {
            const int f = tid + q * 256, g = f / Hp, c = f - g * Hp;
            v[q] = (g < nr && c < D) ? __int_as_float(ld_agent(reinterpret_cast<const int *>(hnew + (size_t)(r0 + g) * Hp + c))) : 0.f;
        }
#pragma unroll
        for (int q = 0; q < CU; ++q) {
            const int f = tid + q * 256, g = f / Hp, c = f - g * Hp;
            if (g < 16 && c < Hp) A[g * lda + Xp + c] = v[q];
            if (g < nr && c < D) memory[(size_t)rid[g] * D + c] = v[q];
        }
    }
    GSTAMP(7);
    if (P == nullptr) return;
    __syncthreads();
    for (int b = wave; b < NT; b += GS_WAVES) {
        const int KCh = Hp / 16;                                         // <= 8
        f32x4 wv[8];
#pragma unroll
        for (int c = 0; c < 8; ++c)
            wv[c] = c < KCh ? *reinterpret_cast<const f32x4 *>(Wm_p + (size_t)(16 * b + r16) * Hp + 16 * c + 4 * g4) : zero4;
        f32x4 acc = zero4;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (c >= KCh) break;
            const f32x4 av = *reinterpret_cast<const f32x4 *>(A + r16 * lda + Xp + 16 * c + 4 * g4);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], wv[c][j], acc, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int g = 4 * g4 + j;
            if (g < nr) P[(size_t)rid[g] * Hp + 16 * b + r16] = acc[j];
        }
    }
}

__global__ __launch_bounds__(64 * GS_WAVES) void k_gru_split(GruSplitArgs GS)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gru_split_body(GS, smem, blockIdx.x, blockIdx.y, nullptr);
}

// k_out_gru for SMALL batches: the latency-organised output layers (k_embed_out2: one wave per (tiles, path, N-tile), four of
// them to a workgroup here) beside k_gru_split.  Workgroup order as in k_out_gru: the source-path waves first (n_src_wgs
// workgroups), the GRU's (tile, N-tile) workgroups, the neighbour paths.
template <int NT, int HG>
__global__ __launch_bounds__(64 * GS_WAVES) void k_out_gru2(EmbedOutArgs E, int gx, int n_src_wgs, int gru_tiles, GruSplitArgs GS, SrcGate gate)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int bid = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int gru_wgs = gru_tiles * NT;
    if (bid >= n_src_wgs && bid < n_src_wgs + gru_wgs) {
        const int g = bid - n_src_wgs;
        gru_split_body(GS, smem, g % gru_tiles, g / gru_tiles, &gate);
        if (threadIdx.x == 0) gate_leave(gate);              // (every GRU workgroup is a participant, whether it waited or not)
        return;
    }
    float *Y = reinterpret_cast<float *>(smem) + wave * (16 * (NT * 16 + 4));
    const int per_path = gx * NT;                          // waves per path: (tile stride gx) x (N-tile)
    if (bid < n_src_wgs) {
        const int v = 4 * bid + wave;
        if (v < per_path) {
            embed_out2_body<NT, HG>(E, Y, lane, v % gx, gx, 0, v / gx, gate.word);
            if (lane == 0) gate_leave(gate);                 // (a source-path wave: lane 0 made its add to word[0])
        }
    } else {
        const int v = 4 * (bid - n_src_wgs - gru_wgs) + wave;
        if (v < per_path * E.M) { const int r = v % per_path; embed_out2_body<NT, HG>(E, Y, lane, r % gx, gx, 1 + v / per_path, r / gx, nullptr); }
    }
}

// (Round 5, measured and removed: k_gru_persist -- k_gru_split with a workgroup's weight fragments kept in registers over a
//  loop of row tiles, the next tile's rows in flight under the current tile's MFMAs.  The split needs a cross-workgroup
//  arrival per tile -- drained sc1 stores, a returning atomic, two barriers -- that a persistent loop pays per TILE where
//  k_gru_split pays it once per workgroup: 8 192 rows 137 us against k_gru's 66, 2 000 rows 47 against 30, 1 200 rows
//  (message 472 wide) 43 against 37; only at 400 rows 25 against 27.  tools/exp/p23_kernels.py, profiles/r5/experiments/.)
constexpr int GP_TILE_COUNTERS = GS_TILE_COUNTERS;
constexpr long long GP_MAX_ROWS = (long long)GS_MAX_ROWS;

struct GruPlan {
    int Xp, Hp, lda;
    size_t lds, off_rows, off_cnt, off_wih, off_whh, off_tiles, off_hnew, total;
};

void gru_plan(int64_t max_rows, int D, int msg_dim, GruPlan &p)
{
    p.Xp = round_up(msg_dim, 16);
    p.Hp = round_up(D, 16);
    p.lda = p.Xp + p.Hp + 4;
    p.lds = (size_t)2 * 16 * p.lda * 4 + 2 * 16 * 4;   // A tile + node ids of k_gru<2> (k_gru<1> uses half)
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 255) & ~(size_t)255; return r; };
    // the packed weights come BEFORE the row list: their place does not depend on max_rows, so a workspace whose
    // weights are packed (weights_ready) stays valid when the next call has a different number of ids
    p.off_cnt = take(256);
    p.off_wih = take((size_t)3 * p.Hp * p.Xp * 4);
    p.off_whh = take((size_t)3 * p.Hp * p.Hp * 4);
    p.off_tiles = take((size_t)GP_TILE_COUNTERS * 4);                    // k_gru_split / k_gru_persist: arrival counters per tile (zeroed with the weights)
    p.off_rows = take((size_t)(max_rows > 0 ? max_rows : 1) * 4);
    // new rows until the tile's commit (k_gru_split, k_gru_persist)
    p.off_hnew = take(max_rows > 0 && max_rows <= GP_MAX_ROWS ? (size_t)((max_rows + 15) / 16 * 16) * p.Hp * 4 : 0);
    p.total = o;
}


// ---- one-node multi-GPU exchange (no reference counterpart) ----
// A touched row travels as float32 [id (int bits) | row of table 0 | row of table 1 | ...].
struct RowTables {
    float *ptr[8];
    int width[8];
    int n, row_floats;
};

// one wavefront per slot r < cap: ids[r] (r < *n_valid) or -1, then the id's row of every table (zeros for -1)
__global__ __launch_bounds__(256) void k_pack_rows(RowTables T, const int *__restrict__ ids, const int *__restrict__ n_valid,
                                                   long long cap, float *__restrict__ out)
{
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= cap) return;
    const int id = r < (long long)*n_valid ? ids[r] : -1;
    float *o = out + r * T.row_floats;
    if (lane == 0) o[0] = __int_as_float(id);
    int col = 1;
    for (int t = 0; t < T.n; ++t) {
        const int w = T.width[t];
        const float *src = T.ptr[t] + (size_t)(id < 0 ? 0 : id) * w;
        for (int c = lane; c < w; c += 64) o[col + c] = id < 0 ? 0.f : src[c];
        col += w;
    }
}

// one wavefront per received row: rows with id >= 0 overwrite the local tables
__global__ __launch_bounds__(256) void k_scatter_rows(RowTables T, const float *__restrict__ recv, long long rows)
{
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    const float *in = recv + r * T.row_floats;
    const int id = __float_as_int(in[0]);
    if (id < 0) return;
    int col = 1;
    for (int t = 0; t < T.n; ++t) {
        const int w = T.width[t];
        float *dst = T.ptr[t] + (size_t)id * w;
        for (int c = lane; c < w; c += 64) dst[c] = in[col + c];
        col += w;
    }
}

static bool make_tables(const zt_row_tables *t, RowTables &T)
{
    if (!t || t->n < 1 || t->n > 8) return false;
    T.n = t->n;
    T.row_floats = 1;
    for (int q = 0; q < t->n; ++q) {
        if (!t->ptr[q] || t->width[q] < 1) return false;
        T.ptr[q] = t->ptr[q]; T.width[q] = t->width[q];
        T.row_floats += t->width[q];
    }
    return true;
}

}  // namespace

extern "C" int zt_store_messages_range(const float *, const float *, const float *, const float *, int64_t, int64_t, int32_t,
                                       int32_t, int32_t, const int32_t *, const int32_t *, const double *,
                                       const int64_t *, int64_t, int64_t, int64_t, float *, float *, uint8_t *,
                                       int32_t *, int32_t *, int32_t *, int32_t *, void *);

extern "C" int zt_store_messages(const float *memory_dev, const float *last_update_dev, const float *efeat_dev,
                                 const float *time_w_dev, int64_t num_nodes, int64_t num_edges, int32_t D, int32_t F,
                                 int32_t T, const int32_t *src_dev, const int32_t *dst_dev, const double *ts_dev,
                                 const int64_t *eidx_dev, int64_t B, float *messages_dev, float *msg_ts_dev,
                                 uint8_t *flags_dev, int32_t *scratch_dev, int32_t *uniq_ids_dev, int32_t *n_uniq_dev,
                                 int32_t *status_dev, void *stream)
{
    return zt_store_messages_range(memory_dev, last_update_dev, efeat_dev, time_w_dev, num_nodes, num_edges, D, F, T,
                                   src_dev, dst_dev, ts_dev, eidx_dev, B, 0, 2 * B, messages_dev, msg_ts_dev, flags_dev,
                                   scratch_dev, uniq_ids_dev, n_uniq_dev, status_dev, stream);
}

extern "C" int zt_store_messages_range(const float *memory_dev, const float *last_update_dev, const float *efeat_dev,
                                       const float *time_w_dev, int64_t num_nodes, int64_t num_edges, int32_t D,
                                       int32_t F, int32_t T, const int32_t *src_dev, const int32_t *dst_dev,
                                       const double *ts_dev, const int64_t *eidx_dev, int64_t B, int64_t pos_lo,
                                       int64_t pos_hi, float *messages_dev, float *msg_ts_dev, uint8_t *flags_dev,
                                       int32_t *scratch_dev, int32_t *uniq_ids_dev, int32_t *n_uniq_dev,
                                       int32_t *status_dev, void *stream)
{
    return zt::store_messages_ex(memory_dev, last_update_dev, efeat_dev, time_w_dev, num_nodes, num_edges, D, F, T, src_dev, dst_dev,
                                 ts_dev, eidx_dev, B, pos_lo, pos_hi, messages_dev, msg_ts_dev, flags_dev, scratch_dev, uniq_ids_dev,
                                 n_uniq_dev, status_dev, nullptr, stream);
}

// ... with one extra: *zero_word_dev = 0 by the first kernel (pipeline.hip: the GRU update's row counter, instead of a
// memset node of its own between the two)
int zt::store_messages_ex(const float *memory_dev, const float *last_update_dev, const float *efeat_dev, const float *time_w_dev,
                          int64_t num_nodes, int64_t num_edges, int32_t D, int32_t F, int32_t T, const int32_t *src_dev,
                          const int32_t *dst_dev, const double *ts_dev, const int64_t *eidx_dev, int64_t B, int64_t pos_lo,
                          int64_t pos_hi, float *messages_dev, float *msg_ts_dev, uint8_t *flags_dev, int32_t *scratch_dev,
                          int32_t *uniq_ids_dev, int32_t *n_uniq_dev, int32_t *status_dev, int32_t *zero_word_dev, void *stream,
                          bool *zeroed_out, bool set_flags)
{
    if (zeroed_out) *zeroed_out = false;
    if (B < 0 || D <= 0 || F < 0 || T < 0 || !status_dev) { set_error("zt_store_messages: bad argument"); return ZT_ERR_ARG; }
    if (B == 0) return ZT_OK;
    if (!memory_dev || !last_update_dev || !efeat_dev || !time_w_dev || !src_dev || !dst_dev || !ts_dev || !eidx_dev ||
        !messages_dev || !msg_ts_dev || !flags_dev || !scratch_dev) {
        set_error("zt_store_messages: NULL buffer");
        return ZT_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    if (n_uniq_dev && n_uniq_dev != zero_word_dev) ZT_HIP(hipMemsetAsync(n_uniq_dev, 0, sizeof(int), s));   // (else k_last_pos zeroes it)
    const long long *e64 = reinterpret_cast<const long long *>(eidx_dev);
    ZT_PROF_BEGIN(s, P_STORE_MSG);
    k_last_pos<<<(unsigned)((2 * B + 255) / 256), 256, 0, s>>>(src_dev, dst_dev, e64, B, num_nodes, num_edges,
                                                               scratch_dev, status_dev, zero_word_dev);
    // two positions per wave where the shape allows (zt_set_kernel_choice(ZT_CHOICE_MESSAGES, ZT_MSG_ONE) pins the other kernel)
    if (zt::kernel_choice(ZT_CHOICE_MESSAGES) != ZT_MSG_ONE && D <= 128 && T <= 128 && F <= 256)
        k_build_messages2<<<(unsigned)(((2 * B + 1) / 2 + 3) / 4), 256, 0, s>>>(
            memory_dev, last_update_dev, efeat_dev, time_w_dev, num_nodes, num_edges, D, F, T, src_dev, dst_dev, ts_dev, e64,
            B, messages_dev, msg_ts_dev, flags_dev, scratch_dev, uniq_ids_dev, n_uniq_dev, status_dev, pos_lo, pos_hi, set_flags ? 1 : 0);
    else
    k_build_messages<<<(unsigned)((2 * B + 3) / 4), 256, 0, s>>>(
        memory_dev, last_update_dev, efeat_dev, time_w_dev, num_nodes, num_edges, D, F, T, src_dev, dst_dev, ts_dev, e64,
        B, messages_dev, msg_ts_dev, flags_dev, scratch_dev, uniq_ids_dev, n_uniq_dev, status_dev, pos_lo, pos_hi, set_flags ? 1 : 0);
    ZT_PROF_END(s, P_STORE_MSG);
    ZT_LAUNCH_CHECK();
    if (zeroed_out) *zeroed_out = zero_word_dev != nullptr;      // k_last_pos ran: the word is zero for whatever follows on this stream
    return ZT_OK;
}

extern "C" int64_t zt_gru_workspace_bytes(int64_t max_rows, int32_t D, int32_t msg_dim)
{
    if (max_rows < 0 || D <= 0 || msg_dim <= 0) return -1;
    GruPlan p;
    gru_plan(max_rows, D, msg_dim, p);
    return (int64_t)p.total;
}

extern "C" int64_t zt_gru_rows_offset(int32_t D, int32_t msg_dim)
{
    if (D <= 0 || msg_dim <= 0) return -1;
    GruPlan p;
    gru_plan(1, D, msg_dim, p);
    return (int64_t)p.off_rows;
}

extern "C" int zt_gru_update(float *memory_dev, float *last_update_dev, const float *messages_dev,
                             const float *msg_ts_dev, uint8_t *flags_dev, int64_t num_nodes, int32_t D,
                             int32_t msg_dim, const int32_t *ids_dev, int64_t n_ids, const int32_t *n_ids_dev,
                             const zt_gru_weights *wt, void *workspace_dev, int32_t weights_ready, void *stream)
{
    return zt::gru_update_ex(memory_dev, last_update_dev, messages_dev, msg_ts_dev, flags_dev, num_nodes, D, msg_dim, ids_dev,
                             n_ids, n_ids_dev, wt, workspace_dev, weights_ready, nullptr, nullptr, stream, false, false);
}

// zt_gru_update with the refresh of the projected table folded in (pipeline.hip): wm_p = W_m padded to [Dp][Dp]
// (zt::embed_wm_ptr), proj_table = [num_nodes][Dp]; both NULL: plain zt_gru_update
int zt::gru_update_ex(float *memory_dev, float *last_update_dev, const float *messages_dev, const float *msg_ts_dev,
                      uint8_t *flags_dev, int64_t num_nodes, int32_t D, int32_t msg_dim, const int32_t *ids_dev, int64_t n_ids,
                      const int32_t *n_ids_dev, const zt_gru_weights *wt, void *workspace_dev, int32_t weights_ready,
                      const float *wm_p, float *proj_table, void *stream, bool counter_zeroed, bool select_done,
                      zt::embed_out_deferred *fuse)
{
    // (output layers held back by embed_ex: launched here whatever happens -- beside the GRU kernel where the shapes allow)
    struct PendingOut {
        zt::embed_out_deferred *d; void *s;
        ~PendingOut() { if (d && d->valid) { (void)zt::embed_out_launch(*d, s); d->valid = false; } }
    } pending{fuse, stream};
    if (!memory_dev || !last_update_dev || !messages_dev || !msg_ts_dev || !flags_dev || !wt || !workspace_dev ||
        D <= 0 || msg_dim <= 0 || n_ids < 0) {
        set_error("zt_gru_update: bad argument");
        return ZT_ERR_ARG;
    }
    if (D > 128) { set_error("zt_gru_update: D=%d > 128 unsupported", D); return ZT_ERR_UNSUPPORTED; }
    const int64_t max_rows = ids_dev ? n_ids : num_nodes;
    if (max_rows == 0) return ZT_OK;
    GruPlan p;
    gru_plan(max_rows, D, msg_dim, p);
    if (p.lds > 150 * 1024) { set_error("zt_gru_update: message width %d too large", msg_dim); return ZT_ERR_UNSUPPORTED; }
    hipStream_t s = (hipStream_t)stream;
    char *ws = reinterpret_cast<char *>(workspace_dev);
    int *cnt = reinterpret_cast<int *>(ws + p.off_cnt);
    int *rows = reinterpret_cast<int *>(ws + p.off_rows);
    float *wih = reinterpret_cast<float *>(ws + p.off_wih);
    float *whh = reinterpret_cast<float *>(ws + p.off_whh);
    if (!counter_zeroed && !select_done) ZT_HIP(hipMemsetAsync(cnt, 0, sizeof(int), s));
    ZT_PROF_BEGIN(s, P_GRU);
    if (!select_done)
        k_select_flagged<<<(unsigned)((max_rows + 255) / 256), 256, 0, s>>>(ids_dev, n_ids, n_ids_dev, num_nodes, flags_dev, rows, cnt);
    if (!weights_ready) {                       // gate-packed, padded copies: once per weight change
        k_pack_gates<<<(3 * p.Hp * p.Xp + 255) / 256, 256, 0, s>>>(wt->w_ih, D, msg_dim, wih, p.Hp, p.Xp);
        k_pack_gates<<<(3 * p.Hp * p.Hp + 255) / 256, 256, 0, s>>>(wt->w_hh, D, D, whh, p.Hp, p.Hp);
        ZT_HIP(hipMemsetAsync(ws + p.off_tiles, 0, (size_t)GP_TILE_COUNTERS * 4, s));
        ZT_HIP(hipMemsetAsync(cnt + GRU_SRC_WORD, 0, 2 * sizeof(int), s));   // the gate's two words (a fresh workspace; afterwards every launch leaves them at zero)
    }
    // Two organisations of the same update (zt_set_kernel_choice(ZT_CHOICE_GRU, ..) pins one; tests hold them against each
    // other and torch's GRUCell):
    //   k_gru_split   <= 512 rows: a workgroup per (16 rows, N-tile) -- the whole chip works on what k_gru gives a tenth of it;
    //   k_gru         beyond: 16 rows per workgroup, the gate weights streamed from L2 per tile.
    const int choice = zt::kernel_choice(ZT_CHOICE_GRU);
    const bool fits = max_rows <= GS_MAX_ROWS && (p.Xp + p.Hp) / 16 <= GS_WAVES * GS_MAXCH && 16 * (msg_dim + D) <= GS_STAGE * 256 &&
                      16 * p.Hp <= 7 * 256;
    const bool split = fits && (choice == ZT_GRU_SPLIT || (choice == 0 && max_rows <= 512));
    if (split) {
        const size_t lds2 = ((size_t)16 * p.lda + (size_t)GS_WAVES * 4 * 64 * 4) * 4 + 32 * 4;
        static size_t attr2 = 0;
        if (lds2 > 48 * 1024 && lds2 > attr2) {
            ZT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gru_split), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
            attr2 = lds2;
        }
        GruSplitArgs GS;
        GS.g.memory = memory_dev; GS.g.last_update = last_update_dev; GS.g.messages = messages_dev; GS.g.msg_ts = msg_ts_dev; GS.g.rows = rows;
        GS.g.n_rows = cnt; GS.g.D = D; GS.g.msg_dim = msg_dim; GS.g.Xp = p.Xp; GS.g.Hp = p.Hp; GS.g.lda = p.lda; GS.g.Wih_p = wih; GS.g.Whh_p = whh;
        GS.g.b_ih = wt->b_ih; GS.g.b_hh = wt->b_hh; GS.g.Wm_p = wm_p; GS.g.P = proj_table; GS.g.cap = (int)max_rows;
        GS.tile_cnt = reinterpret_cast<int *>(ws + p.off_tiles); GS.hnew = reinterpret_cast<float *>(ws + p.off_hnew);
        const int gru_tiles = (int)((max_rows + 15) / 16), NTg = p.Hp / 16;
        const bool can_fuse2 = fuse != nullptr && fuse->valid && fuse->form == 2 && fuse->memory == memory_dev &&
                               (fuse->D + 15) / 16 == NTg && (NTg == 7 || NTg == 8) && (fuse->hg == 1 || fuse->hg == 5 || fuse->hg == 10);
        if (can_fuse2) {
            const zt::embed_out_deferred &d = *fuse;
            EmbedOutArgs E;
            E.memory = d.memory; E.num_nodes = d.num_nodes; E.nodes = d.nodes; E.N = d.N; E.D = d.D; E.M = d.M; E.H = d.H; E.S = d.S;
            E.fc2_p = d.fc2_p; E.fc2_b = d.fc2_b; E.fc1s_p = d.fc1s_p; E.fc1s_b = d.fc1s_b; E.fc2s_p = d.fc2s_p; E.fc2s_b = d.fc2s_b;
            E.out = d.out; E.status = d.status;
            const int per_path = d.gx * NTg, n_src_wgs = (per_path + 3) / 4, n_nb_wgs = (per_path * d.M + 3) / 4;
            size_t lds_f = (size_t)4 * 16 * (NTg * 16 + 4) * 4;
            if (lds_f < lds2) lds_f = lds2;
            SrcGate gate;
            gate.word = cnt + GRU_SRC_WORD; gate.target = (unsigned)per_path; gate.participants = (unsigned)(per_path + gru_tiles * NTg);
            gate.status = d.status; gate.latch = d.latch;
            const unsigned grid = (unsigned)(n_src_wgs + gru_tiles * NTg + n_nb_wgs);
#define ZT_OG2(NTV, HGV) do {                                                                                                   \
                static size_t attr_og2 = 0;                                                                                     \
                if (lds_f > 48 * 1024 && lds_f > attr_og2) {                                                                    \
                    ZT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_out_gru2<NTV, HGV>),                            \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f));                       \
                    attr_og2 = lds_f;                                                                                           \
                }                                                                                                               \
                k_out_gru2<NTV, HGV><<<grid, 64 * GS_WAVES, lds_f, s>>>(E, d.gx, n_src_wgs, gru_tiles, GS, gate);                 \
            } while (0)
            if (NTg == 7) { if (d.hg == 1) ZT_OG2(7, 1); else if (d.hg == 5) ZT_OG2(7, 5); else ZT_OG2(7, 10); }
            else          { if (d.hg == 1) ZT_OG2(8, 1); else if (d.hg == 5) ZT_OG2(8, 5); else ZT_OG2(8, 10); }
#undef ZT_OG2
            fuse->valid = false;
        } else {
            // (held-back output layers first: their source path reads the rows this kernel rewrites)
            if (fuse != nullptr && fuse->valid) { const int rc = zt::embed_out_launch(*fuse, s); fuse->valid = false; if (rc != ZT_OK) return rc; }
            k_gru_split<<<dim3((unsigned)gru_tiles, (unsigned)NTg), 64 * GS_WAVES, lds2, s>>>(GS);
        }
    } else {
        const size_t lds = (size_t)16 * p.lda * 4 + 32 * 4;              // A tile + node ids + the gate's verdict (k_out_gru)
        static size_t attr_lds = 0;
        if (lds > 48 * 1024 && lds > attr_lds) {
            ZT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gru<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_lds = lds;
        }
        GruArgs G;
        G.memory = memory_dev; G.last_update = last_update_dev; G.messages = messages_dev; G.msg_ts = msg_ts_dev; G.rows = rows; G.n_rows = cnt;
        G.D = D; G.msg_dim = msg_dim; G.Xp = p.Xp; G.Hp = p.Hp; G.lda = p.lda; G.Wih_p = wih; G.Whh_p = whh; G.b_ih = wt->b_ih; G.b_hh = wt->b_hh;
        G.Wm_p = wm_p; G.P = proj_table; G.cap = (int)max_rows;
        const unsigned gru_wgs = (unsigned)((max_rows + 15) / 16);
        if (fuse != nullptr && fuse->valid && fuse->form == 1 && fuse->memory == memory_dev && (fuse->hg == 1 || fuse->hg == 5 || fuse->hg == 10)) {
            const zt::embed_out_deferred &d = *fuse;
            EmbedOutArgs E;
            E.memory = d.memory; E.num_nodes = d.num_nodes; E.nodes = d.nodes; E.N = d.N; E.D = d.D; E.M = d.M; E.H = d.H; E.S = d.S;
            E.fc2_p = d.fc2_p; E.fc2_b = d.fc2_b; E.fc1s_p = d.fc1s_p; E.fc1s_b = d.fc1s_b; E.fc2s_p = d.fc2s_p; E.fc2s_b = d.fc2s_b;
            E.out = d.out; E.status = d.status;
            const int out_tiles = (int)((d.N + OUT_ROWS - 1) / OUT_ROWS), n_out = out_tiles * (d.M + 1);
            const int Dp = (d.D + 15) / 16 * 16;
            size_t lds_f = (size_t)2 * OUT_ROWS * (Dp + 4) * 4 + OUT_ROWS * 4;
            if (lds_f < lds) lds_f = lds;
            const void *fn = d.hg == 1 ? reinterpret_cast<const void *>(k_out_gru<1>)
                                       : (d.hg == 5 ? reinterpret_cast<const void *>(k_out_gru<5>) : reinterpret_cast<const void *>(k_out_gru<10>));
            static size_t attr_f[3] = {0, 0, 0};
            const int hi = d.hg == 1 ? 0 : (d.hg == 5 ? 1 : 2);
            if (lds_f > 48 * 1024 && lds_f > attr_f[hi]) {
                ZT_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f));
                attr_f[hi] = lds_f;
            }
            SrcGate gate;
            gate.word = cnt + GRU_SRC_WORD; gate.target = (unsigned)out_tiles; gate.participants = (unsigned)out_tiles + gru_wgs;
            gate.status = d.status; gate.latch = d.latch;
            const unsigned grid = (unsigned)n_out + gru_wgs;
            if (d.hg == 1) k_out_gru<1><<<grid, 64 * GRU_WAVES, lds_f, s>>>(E, out_tiles, (int)gru_wgs, G, gate);
            else if (d.hg == 5) k_out_gru<5><<<grid, 64 * GRU_WAVES, lds_f, s>>>(E, out_tiles, (int)gru_wgs, G, gate);
            else k_out_gru<10><<<grid, 64 * GRU_WAVES, lds_f, s>>>(E, out_tiles, (int)gru_wgs, G, gate);
            fuse->valid = false;                       // (launched)
        } else {
            if (fuse != nullptr && fuse->valid) { const int rc = zt::embed_out_launch(*fuse, s); fuse->valid = false; if (rc != ZT_OK) return rc; }
            k_gru<1><<<gru_wgs, 64 * GRU_WAVES, lds, s>>>(G);
        }
    }
    ZT_PROF_END(s, P_GRU);
    ZT_LAUNCH_CHECK();
    return ZT_OK;
}

#ifdef ZT_GRU_STAMP
extern "C" int zt_debug_gru(unsigned long long *host)
{
    ZT_HIP(hipDeviceSynchronize());
    ZT_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_gru), sizeof(unsigned long long) * 2048 * 10));
    return ZT_OK;
}
#endif

extern "C" int zt_pack_rows(const zt_row_tables *tables, const int32_t *ids_dev, const int32_t *n_valid_dev, int64_t cap,
                            float *out_dev, void *stream)
{
    RowTables T;
    if (!make_tables(tables, T) || !ids_dev || !n_valid_dev || cap < 0 || (cap > 0 && !out_dev)) {
        set_error("zt_pack_rows: bad argument");
        return ZT_ERR_ARG;
    }
    if (cap == 0) return ZT_OK;
    k_pack_rows<<<(unsigned)((cap + 3) / 4), 256, 0, (hipStream_t)stream>>>(T, ids_dev, n_valid_dev, cap, out_dev);
    ZT_LAUNCH_CHECK();
    return ZT_OK;
}

extern "C" int zt_scatter_rows(const zt_row_tables *tables, const float *recv_dev, int64_t rows, void *stream)
{
    RowTables T;
    if (!make_tables(tables, T) || rows < 0 || (rows > 0 && !recv_dev)) {
        set_error("zt_scatter_rows: bad argument");
        return ZT_ERR_ARG;
    }
    if (rows == 0) return ZT_OK;
    k_scatter_rows<<<(unsigned)((rows + 3) / 4), 256, 0, (hipStream_t)stream>>>(T, recv_dev, rows);
    ZT_LAUNCH_CHECK();
    return ZT_OK;
}
