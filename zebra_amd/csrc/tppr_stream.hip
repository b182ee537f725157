// Streaming top-k T-PPR maintenance on MI355X (gfx950).
//
// Replaces tppr_finder.streaming_topk & friends (reference utils/util.py:391-873).
// Compile with -ffp-contract=off: the reference rounds every float64 multiply
// and add separately (utils/util.py:523-541) and results must be bit-exact.
//
// Design (DESIGN.md "P1"):
//   * State is device resident.  A node's dictionary (one per model) is a ROW
//     of 8-byte GRANULES {tag:32 | payload:32}: header (len, norm) and, per
//     entry, key = (edge_idx, node), timestamp, weight split into 32-bit
//     halves, stored as six entry-major arrays so that lane j owns entry j.
//     Every granule of a row carries the tag (launch epoch, writer ordinal) of
//     the edge that wrote it: a row is self-validating, and "the data is the
//     flag" (cdna_hip_programming.md Guideline 16, recipe R2): a dependent edge
//     re-reads the row until all tags match -- ONE memory round trip per hop of
//     a hub chain, no release fence, no acquire, no separate flag.
//   * The edges of a batch must be applied in order, each reading the rows the
//     previous edges wrote.  A prepass groups the batch's node accesses by node
//     (atomic count / reserve / fill) and gives every access (a) the ordinal of
//     the last earlier WRITER of its node = the tag to expect, and (b) if its
//     immediate predecessor on that node was only a reader (a negative
//     sample), that edge's index: writers must not overwrite a row such a
//     reader has not read yet, which is what the per-edge "reads done" flag is
//     for.
//   * The update kernel is persistent: one wavefront per (edge, model) task,
//     tasks dequeued in order from an atomic head (a task only ever waits on
//     tasks that are already resident -> no deadlock); it loads its rows
//     (polling tags where a predecessor exists, with a back-off proportional to
//     the number of chain hops still ahead), publishes "reads done", merges in
//     LDS, prunes with the exact numba argsort semantics, and stores the new
//     rows as write-through (sc1) granules, fire and forget.
//   * The kernel's time is the longest chain of edges through one node times
//     the time of one hop.  The most-touched nodes of a launch (hubs) get a
//     workgroup of their own whose waves take the hub's edges in order and pass
//     the hub's row through an LDS mailbox; a hub row goes to memory only when
//     its next accessor reads it from there (process_edge: hub_to_memory).
//   * The prepass reads only node / edge ids.  The handle keeps two sets of its
//     buffers, so zt_tppr_plan can run it for a later call on another stream
//     while k_stream still works on the previous set.
//   * A merge matches keys through a per-wave hash table in LDS (all pairs on
//     slot collisions); the top-k prune is rank counting in registers, a
//     quicksort replay on the ranks when ties decide, LDS / sequential replays
//     for more than 64 candidates (numba_sort.hpp).
#include "numba_sort.hpp"

#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace zt {

// ---- error plumbing ----------------------------------------------------------
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- kernel timing with HIP events ----------------------------------------------
bool g_prof_on = false;
namespace {
struct ProfRec { hipEvent_t a, b; int id; bool open; };
std::vector<ProfRec> g_prof_recs;
std::vector<hipEvent_t> g_prof_pool;
double g_prof_ms[P_COUNT];
long long g_prof_n[P_COUNT];
const char *const g_prof_names[P_COUNT] = {"tppr_prepass", "tppr_stream", "tppr_cleanup", "pruned_topk",
                                           "embed_prep", "fc1_agg", "embed_out", "store_messages", "gru_update", "score"};
hipEvent_t prof_event()
{
    hipEvent_t e;
    if (!g_prof_pool.empty()) { e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
int g_prof_every = 1;                 // time every n-th launch of each kernel (zt_profile_enable(n))
long long g_prof_seen[P_COUNT];
long long g_prof_open[P_COUNT] = {-1, -1, -1, -1, -1, -1, -1, -1, -1};   // index of the open record per kernel
}  // namespace
void prof_begin(hipStream_t s, int id)
{
    // the T-PPR update is one launch per group of batches on a stream of its own: every launch is timed; of the main
    // stream's kernels every n-th (two event records per kernel are ~3 % of a step there)
    const int every = id <= P_CLEANUP ? 1 : g_prof_every;
    if ((g_prof_seen[id]++ % every) != 0) return;             // not sampled: prof_end finds no open record
    ProfRec r{prof_event(), prof_event(), id, true};
    if (!r.a || !r.b) return;
    (void)hipEventRecord(r.a, s);
    g_prof_open[id] = (long long)g_prof_recs.size();
    g_prof_recs.push_back(r);
}
void prof_end(hipStream_t s, int id)
{
    const long long i = g_prof_open[id];
    if (i < 0 || i >= (long long)g_prof_recs.size() || !g_prof_recs[i].open || g_prof_recs[i].id != id) return;
    (void)hipEventRecord(g_prof_recs[i].b, s);
    g_prof_recs[i].open = false;
    g_prof_open[id] = -1;
}

}  // namespace zt

using namespace zt;

extern "C" int zt_profile_enable(int on)
{
    g_prof_on = on != 0;
    g_prof_every = on > 1 ? on : 1;
    for (int i = 0; i < P_COUNT; ++i) g_prof_seen[i] = 0;
    return ZT_OK;
}

extern "C" int zt_profile_reset(void)
{
    (void)hipDeviceSynchronize();
    for (auto &r : g_prof_recs) { g_prof_pool.push_back(r.a); g_prof_pool.push_back(r.b); }
    g_prof_recs.clear();
    for (int i = 0; i < P_COUNT; ++i) { g_prof_ms[i] = 0; g_prof_n[i] = 0; g_prof_open[i] = -1; }
    return ZT_OK;
}

extern "C" int zt_profile_read(const char *name, int64_t *count, double *total_ms)
{
    if (!name) return ZT_ERR_ARG;
    ZT_HIP(hipDeviceSynchronize());
    for (auto &r : g_prof_recs) {
        if (!r.open) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { g_prof_ms[r.id] += ms; g_prof_n[r.id] += 1; }
        }
        g_prof_pool.push_back(r.a);
        g_prof_pool.push_back(r.b);
    }
    g_prof_recs.clear();
    for (int i = 0; i < P_COUNT; ++i) g_prof_open[i] = -1;
    for (int i = 0; i < P_COUNT; ++i)
        if (strcmp(name, g_prof_names[i]) == 0) {
            if (count) *count = g_prof_n[i];
            if (total_ms) *total_ms = g_prof_ms[i];
            return ZT_OK;
        }
    set_error("zt_profile_read: unknown kernel name %s", name);
    return ZT_ERR_ARG;
}

extern "C" const char *zt_last_error(void) { return g_err; }
extern "C" const char *zt_version(void) { return "zebra_amd 0.1 gfx950"; }

// ---- handle --------------------------------------------------------------------
namespace {
constexpr int HDR = 4;                 // header granules: len, norm lo, norm hi, pad
constexpr int ORD_BITS = 15;           // writer ordinal inside one launch (a node has at most MAX_CHUNK writers)
constexpr int MAX_CHUNK = 16384;       // edges per launch (ordinals must fit ORD_BITS)
constexpr unsigned EPOCH_MAX = (1u << (32 - ORD_BITS)) - 1;
static_assert(MAX_CHUNK == zt::TPPR_MAX_LAUNCH, "common.hpp: TPPR_MAX_LAUNCH");
// hub chains: the nodes touched most often in a launch get a workgroup of their own
constexpr int HOT_MIN = 24;            // accesses in one launch that make a node a chain candidate
constexpr int MAX_HOT = 128;           // candidates kept
constexpr int MAX_CHAINS = 16;         // chains per model
constexpr int CTL_LOG = 6;             // timeout reports kept per launch
constexpr int CTL_WORDS = 16 + 8 * CTL_LOG;
constexpr int CH_MAX = 2048;           // edges per chain (the rest go through the general queue)
}  // namespace

struct zt_tppr {
    int64_t N;
    int32_t k, M;
    int32_t rg;      // granules per row = HDR + 6k
    double alpha[16], beta[16];
    u64 *rows;       // [M][N][rg] granules
    // per-node prepass scratch (device)
    int *cnt;        // [N], zero between calls
    int *off;        // [N]
    // per-launch scratch (device), sized for 3*MAX_CHUNK accesses
    int *slot;       // position of the access inside its node group, -1 invalid, -2 shadow
    int *list;       // accesses grouped by node
    int *wo;         // ordinal of the last earlier writer of the access' node (0 = none in this launch)
    int *pflag;      // edge whose "reads done" flag must be seen first, or -1
    int *nxt;        // number of later edges of the launch touching the access' node (chain still ahead)
    unsigned *done;  // [M][MAX_CHUNK] reads-done flag per (model, edge) = epoch
    unsigned *cdone; // [M][MAX_CHUNK] the same for the CHAIN's reads of a chain-owned edge (hub + partner row)
    u64 *hubver;     // [M][MAX_CHAINS][CH_MAX + 1][rg]: version t of a chain's hub row = the row before chain position t
                     // (dictionary order, tagged with the launch epoch), or nullptr (k > REG_K_MAX: no chains)
    // hub chains of the launch
    int *chain_of;     // [N] chain index of a hub node, -1 otherwise (all -1 between calls)
    int *hot_node;     // [MAX_HOT] candidates, hot_cnt their access counts
    int *hot_cnt;
    int *chain_node;   // [MAX_CHAINS]
    int *chain_len;    // [MAX_CHAINS]
    int *chain_edges;  // [MAX_CHAINS][CH_MAX] edges owned by the chain, ascending
    int *owner_of;     // [MAX_CHUNK] chain owning the edge, or -1
    int *pos_of;       // [MAX_CHUNK] its position in that chain's edge list
    // control words (device): [0] cursor, [1] queue head, [2] status, [3] hot candidates, [4] chains,
    // [13] timeout reports, [16..] the reports (see note_timeout)
    int *ctl;
    unsigned epoch;
    int n_cu;
    int run_cus;     // CUs of the stream the last k_stream ran on (0: not known yet)
    int wg_per_cu;   // k_stream workgroups one CU can hold (hipOccupancyMaxActiveBlocksPerMultiprocessor)
    // Failure latch in host-mapped memory: the first ZT_ERR_RANGE / ZT_ERR_TIMEOUT of any launch is written
    // here by the device (system scope), so the NEXT host call on the handle fails without a synchronisation
    // even when the caller never polls zt_tppr_status.  Cleared by zt_tppr_status.
    int *latch_host, *latch_dev;
    unsigned long long plan_serial;   // tokens handed out by zt_tppr_plan
    // Two sets of the prepass buffers above (the fields above point into the set in use): the prepass of
    // the next call can run on another stream while k_stream still reads the previous call's set.
    struct PlanSet {
        int *cnt, *off, *slot, *list, *wo, *pflag, *nxt, *chain_of, *hot_node, *hot_cnt, *chain_node, *chain_len,
            *chain_edges, *owner_of, *pos_of, *ctl;
        hipEvent_t planned, consumed;      // prepass finished / k_stream finished with the set
        bool used;                         // `consumed` has been recorded at least once
        // what the set was planned for (valid == a zt_tppr_plan result not consumed yet)
        bool valid;
        const int32_t *nodes;
        int B, n_roles, model, grid, max_chains;
        unsigned long long token;
    } set[2];
    int next_set;
    // last launch (diagnostics)
    const int *dbg_nodes;
    long long dbg_stride;
    int dbg_B, dbg_roles, dbg_models;
};

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
#else
#define STAMP(slot) do { } while (0)
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

__host__ __device__ inline u64 granule(unsigned tag, unsigned payload) { return ((u64)tag << 32) | payload; }

// ---------------------------------------------------------------- prepass ----
// One access = (role r, edge i) of the chunk; a = r*B + i.  Roles 0/1 (source,
// destination) write their node's row, role 2 (negative) only reads it.  An
// edge's second access to the same node (self-loop, negative == endpoint) is a
// SHADOW: it is not entered into the node's group.
// K1: validate ids; count accesses per node; remember each access' slot.
__device__ __forceinline__ void latch_failure(int *latch, int code)
{
    __hip_atomic_store(latch, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// K0: a set is planned afresh: the status of the launch it last served is history (the handle's latch keeps it)
__global__ void k_plan_begin(int *ctl) { if (threadIdx.x == 0) { ctl[2] = 0; ctl[13] = 0; } }

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
        slot[a] = -1;
        return;
    }
    bool shadow = false;
    if (r >= 1) shadow = nodes[i] == x;                                    // same as the source
    if (r == 2) shadow = shadow || nodes[role_stride + i] == x;           // same as the destination
    if (shadow) { slot[a] = -2; return; }
    slot[a] = atomicAdd(&cnt[x], 1);
}

__global__ void k_count(const int *__restrict__ nodes, const long long *__restrict__ eidx, long long role_stride,
                        int B, int n_roles, long long N, int *cnt, int *slot, int *ctl, int *latch)
{
    d_count(blockIdx.x * blockDim.x + threadIdx.x, nodes, eidx, role_stride, B, n_roles, N, cnt, slot, ctl, latch);
}

// K2: the first access of each node reserves a contiguous range of `list`.
__device__ __forceinline__ void d_reserve(int a, const int *__restrict__ nodes, long long role_stride, int B, int n_roles,
                                          const int *cnt, int *off, const int *slot, int *ctl, int *hot_node, int *hot_cnt)
{
    if (a >= B * n_roles) return;
    if (slot[a] == 0) {
        const int x = nodes[(long long)(a / B) * role_stride + a % B];
        const int c = cnt[x];
        off[x] = atomicAdd(&ctl[0], c);
        if (c >= HOT_MIN) {                        // hub candidate
            const int hi = atomicAdd(&ctl[3], 1);
            if (hi < MAX_HOT) { hot_node[hi] = x; hot_cnt[hi] = c; }
        }
    }
}

__global__ void k_reserve(const int *__restrict__ nodes, long long role_stride, int B, int n_roles, const int *cnt,
                          int *off, const int *slot, int *ctl, int *hot_node, int *hot_cnt)
{
    d_reserve(blockIdx.x * blockDim.x + threadIdx.x, nodes, role_stride, B, n_roles, cnt, off, slot, ctl, hot_node, hot_cnt);
}

// K2b (one wavefront pair): keep the MAX_CHAINS most-touched candidates as chains.
__device__ __forceinline__ void d_hot_select(int t, int *ctl, const int *hot_node, const int *hot_cnt, int *chain_of,
                                             int *chain_node, int *chain_len, int max_chains)
{
    if (t >= MAX_HOT) return;                      // MAX_HOT threads take part
    int nh = ctl[3];
    nh = nh < MAX_HOT ? nh : MAX_HOT;
    if (t < MAX_CHAINS) chain_len[t] = 0;
    if (ctl[2] == ZT_ERR_RANGE) { if (t == 0) ctl[4] = 0; return; }
    int rank = 0;
    if (t < nh) {
        const int c = hot_cnt[t], x = hot_node[t];
        for (int q = 0; q < nh; ++q) {
            const int cq = hot_cnt[q];
            rank += (cq > c || (cq == c && hot_node[q] < x)) ? 1 : 0;
        }
        if (rank < max_chains) { chain_of[x] = rank; chain_node[rank] = x; }
    }
    if (t == 0) ctl[4] = nh < max_chains ? nh : max_chains;
}

__global__ void k_hot_select(int *ctl, const int *hot_node, const int *hot_cnt, int *chain_of, int *chain_node,
                             int *chain_len, int max_chains)
{
    d_hot_select(threadIdx.x, ctl, hot_node, hot_cnt, chain_of, chain_node, chain_len, max_chains);
}

// K2c: every edge with a hub endpoint is owned by that hub's chain (the more-touched hub if both are).
__device__ __forceinline__ void d_own(int i, const int *__restrict__ nodes, long long role_stride, int B, const int *cnt,
                                      const int *slot, const int *chain_of, int *chain_len, int *chain_edges, int *owner_of)
{
    if (i >= B) return;
    int owner = -1;
    if (slot[i] >= 0 && slot[B + i] != -1) {        // valid edge
        const int u = nodes[i], v = nodes[role_stride + i];
        const int cu = chain_of[u], cv = chain_of[v];
        if (cu >= 0 && (cv < 0 || cnt[u] >= cnt[v])) owner = cu;
        else if (cv >= 0) owner = cv;
        if (owner >= 0) {
            const int p = atomicAdd(&chain_len[owner], 1);
            if (p < CH_MAX) chain_edges[owner * CH_MAX + p] = i;
            else owner = -1;                        // chain full: the general queue takes it
        }
    }
    owner_of[i] = owner;
}

__global__ void k_own(const int *__restrict__ nodes, long long role_stride, int B, const int *cnt, const int *slot,
                      const int *chain_of, int *chain_len, int *chain_edges, int *owner_of)
{
    d_own(blockIdx.x * blockDim.x + threadIdx.x, nodes, role_stride, B, cnt, slot, chain_of, chain_len, chain_edges, owner_of);
}

// K2d: one wavefront per chain sorts its edges ascending (rank by counting).
__global__ __launch_bounds__(64) void k_chain_sort(const int *ctl, const int *chain_len, int *chain_edges, int *pos_of)
{
    __shared__ int e[CH_MAX];
    const int c = blockIdx.x, lane = threadIdx.x;
    if (c >= ctl[4]) return;
    int len = chain_len[c];
    len = len < CH_MAX ? len : CH_MAX;
    for (int p = lane; p < len; p += 64) e[p] = chain_edges[c * CH_MAX + p];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int p = lane; p < len; p += 64) {
        const int me = e[p];
        int r = 0;
        for (int q = 0; q < len; ++q) r += e[q] < me ? 1 : 0;
        chain_edges[c * CH_MAX + r] = me;           // edge indices are distinct
        pos_of[me] = r;
    }
}

// K3: scatter accesses into their node's range, encoded (edge << 2) | role.
__device__ __forceinline__ void d_fill(int a, const int *__restrict__ nodes, long long role_stride, int B, int n_roles,
                                       const int *off, const int *slot, int *list)
{
    if (a >= B * n_roles) return;
    const int s = slot[a];
    if (s < 0) return;
    list[off[nodes[(long long)(a / B) * role_stride + a % B]] + s] = ((a % B) << 2) | (a / B);   // (edge << 2) | role
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
__device__ __forceinline__ void d_deps(int a, const int *__restrict__ nodes, long long role_stride, int B, int n_roles,
                                       const int *cnt, const int *off, const int *slot, const int *list, int *wo,
                                       int *pflag, int *nxt)
{
    if (a >= B * n_roles) return;
    if (slot[a] < 0) { wo[a] = 0; pflag[a] = -1; nxt[a] = 0; return; }
    const int x = nodes[(long long)(a / B) * role_stride + a % B];
    const int o = off[x], c = cnt[x];
    const int me = a % B;
    int best = -1, best_role = 0, writers = 0, nx = 0;
    // 16 list entries are fetched before any is used: a hub's group has hundreds of members and the
    // loop is otherwise one L2 round trip per entry
    for (int p0 = 0; p0 < c; p0 += 16) {
        int b[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) b[t] = list[o + ((p0 + t) < c ? (p0 + t) : (c - 1))];
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
    wo[a] = writers;
    nxt[a] = nx;
    pflag[a] = (best >= 0 && best_role == 2) ? best : -1;
}

__global__ void k_deps(const int *__restrict__ nodes, long long role_stride, int B, int n_roles, const int *cnt,
                       const int *off, const int *slot, const int *list, int *wo, int *pflag, int *nxt)
{
    d_deps(blockIdx.x * blockDim.x + threadIdx.x, nodes, role_stride, B, n_roles, cnt, off, slot, list, wo, pflag, nxt);
}

// K5: restore the per-node counters and the control words for the next call.
__device__ __forceinline__ void d_cleanup(int a, const int *__restrict__ nodes, long long role_stride, int B, int n_roles,
                                          const int *slot, int *cnt, int *ctl, const int *hot_node, int *chain_of)
{
    if (a < MAX_HOT && a < ctl[3]) chain_of[hot_node[a]] = -1;
    if (a >= B * n_roles) return;
    if (slot[a] == 0) cnt[nodes[(long long)(a / B) * role_stride + a % B]] = 0;
}

__global__ void k_cleanup(const int *__restrict__ nodes, long long role_stride, int B, int n_roles, const int *slot,
                          int *cnt, int *ctl, const int *hot_node, int *chain_of)
{
    d_cleanup(blockIdx.x * blockDim.x + threadIdx.x, nodes, role_stride, B, n_roles, slot, cnt, ctl, hot_node, chain_of);
}

// K6: control words back to zero (after k_cleanup has read ctl[3]); ctl[4] = chains stays for k_stream.
__global__ void k_reset_ctl(int *ctl) { if (threadIdx.x < 4 && threadIdx.x != 2) ctl[threadIdx.x] = 0; }

// The whole prepass as ONE workgroup, for launches of at most PRE_FUSED_MAX accesses (small batches: there the ten
// launches above cost more on the host and in launch gaps than the work itself).  Same steps, same arrays, same
// results; the steps are separated by workgroup barriers instead of kernel boundaries.
constexpr int PRE_FUSED_MAX = 12288;
constexpr int PRE_THREADS = 1024;
__global__ __launch_bounds__(PRE_THREADS) void k_prepass_fused(
    const int *__restrict__ nodes, const long long *__restrict__ eidx, long long role_stride, int B, int n_roles,
    long long N, int *cnt, int *slot, int *off, int *list, int *wo, int *pflag, int *nxt, int *ctl, int *latch,
    int *hot_node, int *hot_cnt, int *chain_of, int *chain_node, int *chain_len, int *chain_edges, int *owner_of,
    int *pos_of, int max_chains)
{
    __shared__ int e[CH_MAX];
    const int tid = threadIdx.x, A = B * n_roles;
    if (tid == 0) { ctl[2] = 0; ctl[13] = 0; }                                   // k_plan_begin
    __syncthreads();
    for (int a = tid; a < A; a += PRE_THREADS) d_count(a, nodes, eidx, role_stride, B, n_roles, N, cnt, slot, ctl, latch);
    __syncthreads();
    for (int a = tid; a < A; a += PRE_THREADS) d_reserve(a, nodes, role_stride, B, n_roles, cnt, off, slot, ctl, hot_node, hot_cnt);
    __syncthreads();
    for (int a = tid; a < A; a += PRE_THREADS) d_fill(a, nodes, role_stride, B, n_roles, off, slot, list);
    __syncthreads();
    for (int a = tid; a < A; a += PRE_THREADS) d_deps(a, nodes, role_stride, B, n_roles, cnt, off, slot, list, wo, pflag, nxt);
    d_hot_select(tid, ctl, hot_node, hot_cnt, chain_of, chain_node, chain_len, max_chains);
    __syncthreads();
    for (int i = tid; i < B; i += PRE_THREADS) d_own(i, nodes, role_stride, B, cnt, slot, chain_of, chain_len, chain_edges, owner_of);
    __syncthreads();
    const int n_ch = max_chains > 0 ? ctl[4] : 0;                                // k_chain_sort, chain after chain
    for (int c = 0; c < n_ch; ++c) {
        int len = chain_len[c];
        len = len < CH_MAX ? len : CH_MAX;
        for (int p = tid; p < len; p += PRE_THREADS) e[p] = chain_edges[c * CH_MAX + p];
        __syncthreads();
        for (int p = tid; p < len; p += PRE_THREADS) {
            const int me = e[p];
            int r = 0;
            for (int q = 0; q < len; ++q) r += e[q] < me ? 1 : 0;
            chain_edges[c * CH_MAX + r] = me;
            pos_of[me] = r;
        }
        __syncthreads();
    }
    for (int a = tid; a < (A > MAX_HOT ? A : MAX_HOT); a += PRE_THREADS)
        d_cleanup(a, nodes, role_stride, B, n_roles, slot, cnt, ctl, hot_node, chain_of);
    __syncthreads();
    if (tid < 4 && tid != 2) ctl[tid] = 0;                                       // k_reset_ctl
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
    if (lane >= k) return;
    if (r.len == 0) { on[lane] = 0; oe[lane] = 0; od[lane] = 0.f; ow[lane] = 0.f; return; }
    const bool a = lane < r.len;
    on[lane] = a ? (int)(unsigned)(r.key & 0xffffffffull) : 0;
    oe[lane] = a ? (int)(unsigned)(r.key >> 32) : 0;
    ow[lane] = a ? (float)r.w : 0.f;
    const float tsf = a ? (float)r.ts : 0.f;      // tmp_timestamps is float32
    od[lane] = (float)(tnow - (double)tsf);        // f64 - f32 -> f64 -> stored f32
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
    for (int q = 0; q < nb; ++q) {
        const double x = readlane_f64(bw, q);
        rb += (x > bw || (x == bw && q < lane)) ? 1 : 0;
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
__device__ inline bool load_row_wait(const zt_tppr &h, int m, long long x, int lane, unsigned expect, Row &r,
                                     int *status, unsigned *last_seen = nullptr)
{
    unsigned polls = 0;
    long long t0 = 0;
    for (;;) {
        const unsigned seen = load_row(h, m, x, lane, expect, r);
        if (last_seen) *last_seen = seen;
        if (seen == expect) return true;
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
        if ((++polls & 255u) == 0) {
            const long long now = (long long)wall_clock64();
            if (t0 == 0) t0 = now;
            else if (now - t0 > WAIT_TICKS) { note_timeout(status, 2, (int)x, (int)expect, (int)seen, m); return false; }
            if (launch_failed(status)) return false;
        }
    }
}

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
constexpr int MAIL_R = WAVES_PER_WG;
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
struct Mail {
#ifdef ZT_CRIT
    long long t_start; // core clock when the workgroup started (diagnostic)
#endif
    MailSlot slot[MAIL_R];
    int head;          // next position of the chain's edge list
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
    int sub_B;         // > 0: the launch covers several consecutive batches of sub_B edges (the last may be shorter); the
                       // output rows of batch g form their own [n_models][n_roles][B_g][k] block, blocks back to back
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

__device__ inline void process_edge(const zt_tppr &h, const StreamArgs &A, WaveLds &L, int lane, int i, int mo, Mail *mail,
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
    WL(0, 1); WL(1, blockIdx.x * WAVES_PER_WG + threadIdx.x / WAVE); WL(5, mail ? prev_edge : -2); WL(2, wall_clock64() >> 7);
    int wl_fail = 0;
    unsigned wl_seen = 0;

    // ---- dependencies of this edge's three accesses ----
    int my_wo = 0, my_pf = -1, my_nx = 0;
    if (lane < n_roles) { my_wo = h.wo[lane * B + i]; my_pf = h.pflag[lane * B + i]; my_nx = h.nxt[lane * B + i]; }
    if (my_pf >= 0 && !wait_flag(done + my_pf, epoch, h.ctl + 2, my_pf)) wl_fail |= 1;   // a reader before me has not read yet
    wl_fail = __ballot(wl_fail != 0) != 0ull ? 1 : 0;
    WL(0, 2);
    const int wo_u = __shfl(my_wo, 0), wo_v = __shfl(my_wo, 1), wo_g = __shfl(my_wo, 2);
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

    // Must the hub's new row also go to memory?  Not when the chain's next edge takes it from the
    // mailbox and nobody reads it in between (no reader precedes that edge's access): the next edge's
    // own row supersedes it.  This is a correctness rule, not only a saving: the mailbox hand-off is
    // NOT ordered against this wave's row stores, so a successor could otherwise get its (newer) row
    // into memory before ours and ours would then overwrite it.  Whenever the row IS stored, its next
    // accessor waits for it in memory (directly, or through a reader's reads-done flag).
    bool hub_to_memory = true;
    // next_by_mail: the chain's next edge takes this hop's new row (nobody else writes the hub in between): that row is
    // then also the next position's VERSION (hub_version), which its partner task reads; otherwise the next hop stores
    // its version itself, from the row it finds in memory
    bool next_by_mail = false;
    if (mail != nullptr && next_edge >= 0) {
        const int nrole = A.nodes[next_edge] == hub ? 0 : 1;
        const int my_out = ((u == hub) ? wo_u : wo_v) + 1;
        next_by_mail = h.wo[nrole * B + next_edge] == my_out;
        hub_to_memory = !(next_by_mail && h.pflag[nrole * B + next_edge] < 0);
    }

    // ---- rows: one memory round trip; poll where a writer of this launch precedes us ----
    Row ru, rv, rg;
    const bool u_mail = hub_by_mail && u == hub, v_mail = hub_by_mail && v == hub && v != u;
    unsigned su = 0, sv = 0, sg = 0;
    if (!u_mail) su = load_row(h, m, u, lane, wo_u ? (tag_base | (unsigned)wo_u) : 0u, ru);
    if (v != u && !v_mail) sv = load_row(h, m, v, lane, wo_v ? (tag_base | (unsigned)wo_v) : 0u, rv);
    // (a chain wave applies the HUB's update only: the partner's update and the emission of this edge's rows are a
    //  general task of their own, process_chain_partner -- the negative sample's row is not needed here)
    const bool g_own = mail == nullptr && n_roles == 3 && g != u && g != v;
    if (g_own) sg = load_row(h, m, g, lane, wo_g ? (tag_base | (unsigned)wo_g) : 0u, rg);
    WL(0, 3);
    if (!u_mail && wo_u && su != (tag_base | (unsigned)wo_u))
        if (!load_row_wait(h, m, u, lane, tag_base | (unsigned)wo_u, ru, h.ctl + 2, &wl_seen)) wl_fail |= 2;
    WL(0, 4);
    if (v != u && !v_mail && wo_v && sv != (tag_base | (unsigned)wo_v))
        if (!load_row_wait(h, m, v, lane, tag_base | (unsigned)wo_v, rv, h.ctl + 2, &wl_seen)) wl_fail |= 4;
    WL(0, 5);
    if (g_own && wo_g && sg != (tag_base | (unsigned)wo_g))
        if (!load_row_wait(h, m, g, lane, tag_base | (unsigned)wo_g, rg, h.ctl + 2, &wl_seen)) wl_fail |= 8;
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
                __builtin_amdgcn_s_setprio(0);
            }
            if (hub_to_memory || xa != hub) store_row_scatter(h, m, xa, lane, n, c, new_norm, tag_base | (unsigned)(oa + 1));
            if (mail != nullptr && xa == hub && next_by_mail) store_row_scatter_at(hub_version(h, m, chain_idx, tpos + 1), k, lane, n, c, new_norm, vtag);
        };
        if (!split && mail != nullptr) {
            // a chain wave whose hub row did not come through the mailbox in set order (first hop, self-loop, the partner is
            // the busier node): the hub's update in one piece; the old row is this position's version if nobody stored it
            if (x1 == hub) update(r1, r2, x1, x2, o1, pre_hash, mo == 0 ? i : -1);
            else update(r2, r1, x2, x1, o2, 0, -1);
            if (!hub_by_mail) {
                const Row &ro = hub_is_u ? ru : rv;
                store_row_at(hub_version(h, m, chain_idx, tpos), k, lane, ro.len, ro.key, ro.ts, ro.w, ro.norm, vtag);
            }
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
            if (next_by_mail) store_row_scatter_at(hub_version(h, m, chain_idx, tpos + 1), k, lane, n_new, c, new_norm, vtag);
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
    WL(4, wall_clock64() >> 7); WL(0, 9);
    (void)wl_fail;
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
    int wo_h, wo_p;    // ordinal of the last earlier writer of the hub / of the partner (the tags to expect)
    int pf_h;          // a reader of the hub's row that must be done before this hop may store it (-1: none)
    int wo_prev;       // the same ordinal at the chain's previous edge
    int wo_next, pf_next;   // ... and at its next edge (-1: there is none)
};

__device__ inline bool chain_hop(const zt_tppr &h, const StreamArgs &A, WaveLds &L, int lane, int i, int mo, Mail *mail,
                                 long long hub, int prev_edge, int next_edge, int tpos, ChainHint *hint, int chain_idx,
                                 const HopRec &rec)
{
    if (prev_edge < 0 || h.k > REG_K_MAX) return false;
    const int k = h.k;
    const int m = A.m_lo + mo;
    if (rec.partner < 0) return false;
    const long long pnode = rec.partner;
    const int wo_h = rec.wo_h, wo_p = rec.wo_p;
    if (rec.wo_prev + 1 != wo_h) return false;                                // somebody else wrote the hub in between
    const double alpha = h.alpha[m], beta = h.beta[m];
    unsigned *done = h.done + (long long)m * MAX_CHUNK;
    const unsigned epoch = A.epoch, tag_base = epoch << ORD_BITS, vtag = tag_base | 1u;
#ifdef ZT_CRIT
    long long crit_t[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    int wl_fail = 0;
    CRIT(4);
    // a reader of the hub's row in memory that precedes this edge must have read it before this hop may store there
    // ---- the partner's row from memory (poll where a writer of this launch precedes us) ----
    Row rp;
    const unsigned ptag = wo_p ? (tag_base | (unsigned)wo_p) : 0u;
    RawRow praw;
    load_row_issue(h.rows + ((long long)m * h.N + pnode) * h.rg, k, lane, praw);     // on its way while the rest is looked up
    const double tnow = A.tsv[i];
    const long long e = A.eidx[i];
    const int pf_h = rec.pf_h;
    if (pf_h >= 0 && !wait_flag(done + pf_h, epoch, h.ctl + 2, pf_h)) wl_fail |= 1;
    bool hub_to_memory = true, next_by_mail = false;                        // see process_edge
    if (next_edge >= 0) {
        next_by_mail = rec.wo_next == wo_h + 1;
        hub_to_memory = !(next_by_mail && rec.pf_next < 0);
    }
    MailSlot *in_slot = &mail->slot[(tpos - 1) % MAIL_R], *out_slot = &mail->slot[tpos % MAIL_R];
    // (the two float64 divisions of the scale factors: while the partner's row is on its way)
    PreScale pre_scale;
    pre_scale.valid = false;
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
    if (row_from_raw(praw, k, lane, ptag, rp) != ptag)
        if (!load_row_wait(h, m, pnode, lane, ptag, rp, h.ctl + 2)) wl_fail |= 2;
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
    PreB pre_b;
    pre_b.ok = false;
    if (pre_hash != 2) prepare_b(lane, k, alpha, rp, nkey, tnow, pre_scale, pre_b, h2slot);
    // the partner's row has arrived (its tags were looked at): the partner task may store the partner's new row.  (No
    // s_waitcnt vmcnt(0) here: it would also wait for the write-through stores of this wave's previous hop.)
    st_agent(h.cdone + (long long)m * MAX_CHUNK + i, epoch);
    CRIT(5);
#ifdef ZT_CRIT
    crit_t[9] = !pre_scale.valid ? 7 : (pre_hash == 2 ? 8 : (!pre_b.ok ? 9 : 0));     // why the partner's side is not prepared
    crit_t[13] = (long long)ld_agent(h.ctl + 1) * 100000 + i;    // head of the general queue (task index) when this hop was ready, and its edge
#endif
    // waves whose turn is two or more hops away doze (see process_edge)
    if (tpos >= 2) {
        const int *far = &mail->slot[(tpos - 2) % MAIL_R].seq_set;
        unsigned spins = 0;
        while (lds_load_seq(far) != tpos - 1 && lds_load_seq(&in_slot->seq_set) != tpos) {
            __builtin_amdgcn_s_sleep(8);
            if ((++spins & 1023u) == 0 && launch_failed(h.ctl + 2)) break;
        }
    }
    // (a wave's first hop of a launch has no norm to start from until somebody has published: the kept set two positions
    //  back is out now -- its successor is in its critical section --, which leaves time to prepare the partner's side)
    if (!pre_scale.valid && tpos >= 2 && pre_hash != 2) {
        const MailSlot *sl = &mail->slot[(tpos - 2) % MAIL_R];
        if (lds_load_seq(&sl->seq_set) == tpos - 1) {
            double hn; int a0, a1, a2, a3; unsigned a4;
            mail_hdr_read(sl, hn, a0, a4, a1, a2, a3);
            pn = hn; psteps = 1;
            set_scale();
            if (pre_scale.valid) prepare_b(lane, k, alpha, rp, nkey, tnow, pre_scale, pre_b, h2slot);
        }
    }
    if (!wait_seq(&in_slot->seq_set, tpos, h.ctl + 2, i, prev_edge, true)) wl_fail |= 16;
    __builtin_amdgcn_s_setprio(3);
    CRIT(0);
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
    if (pre_b.ok) {
        lean_done = [&]() -> bool {
            double hn;
            int hlen_v, hmunc_v, hnalt_v, hsorted_v;
            unsigned hunc_v;
            mail_hdr_read(in_slot, hn, hlen_v, hunc_v, hmunc_v, hnalt_v, hsorted_v);
            const bool low = __builtin_amdgcn_inverse_ballot_w64(0xffffffffull);        // lanes 0..31: the hub's entries
            u64 ckey = pre_b.cb_key;
            double cts = pre_b.cb_ts, cw = pre_b.cb_w, hw = 0.0;
            if (low) { ckey = in_slot->key[lane]; cts = in_slot->ts[lane]; hw = in_slot->w[lane]; }
            const int fs = lds_load_seq(&out_slot->seq_free);
            const int n1 = __builtin_amdgcn_readfirstlane(hlen_v), munc = __builtin_amdgcn_readfirstlane(hmunc_v);
            const int nalt = __builtin_amdgcn_readfirstlane(hnalt_v);
            const unsigned hunc = (unsigned)__builtin_amdgcn_readfirstlane((int)hunc_v);
            {   // sorted arrangement, predicted norm (bit patterns on the scalar unit: both are finite and positive)
                const long long hb = __double_as_longlong(hn), pb = __double_as_longlong(pre_scale.norm);
                const unsigned h0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)hb), h1 = (unsigned)__builtin_amdgcn_readfirstlane((int)(hb >> 32));
                const unsigned p0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)pb), p1 = (unsigned)__builtin_amdgcn_readfirstlane((int)(pb >> 32));
                if (__builtin_amdgcn_readfirstlane(hsorted_v) == 0 || h0 != p0 || h1 != p1 || (h0 | h1) == 0u) { LEANC(1); return false; }
            }
            const int nb = pre_b.nb, n = n1 + nb, drop = n - k;
            if (n1 <= 0 || drop <= 0) { LEANC(2); return false; }
            const bool table = lenp > 0;
            if (munc > 0) {
                // members of a straddling run that were not picked may turn out to be in the row (see below)
                const u64 alt = in_slot->alt_key[lane & 31];
                bool t = lane < nalt && alt == nkey;
                if (table) t = t || (lane < nalt && L.htab[key_hash_by(alt, pre_hash)] >= 0);
                if (__ballot(t) != 0ull) { LEANC(3); return false; }
            }
            // is a key of the hub's row in the partner's row?  Read now, looked at after the network
            const bool in1 = lane < n1;
            const int cand = (table && in1) ? L.htab[key_hash_by(ckey, pre_hash)] : -1;
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
            const u64 S = __ballot(__longlong_as_double(((long long)lhi << 32) | (unsigned)llo) != sw) | 1ull;     // run starts
            const u64 below = S & (((u64)2 << lane) - 1ull);
            const int rs = 63 - __builtin_clzll(below);
            const int both = push_i32(rs | (lane << 8), sid);   // (smaller candidates, sorted position) to the candidate's lane
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
            CRIT(1);
            const int lt = both & 0xff, sp = both >> 8;
            const bool full = (S >> drop) & 1ull;               // the cut falls on a run start: exactly k candidates are kept
            const unsigned kmask = (1u << k) - 1u;
            const unsigned claimed = (unsigned)(S >> drop) & kmask;
            const int mode = full ? (claimed == kmask ? FR_RANKS : FR_TIES) : FR_STRADDLE;
            const u64 lowdrop = ((u64)2 << drop) - 1ull;        // positions 0 .. drop
            const int rsG = 63 - __builtin_clzll(S & lowdrop);  // start of the run that holds position `drop`
            if (munc > 0) {
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
            if (tpos >= MAIL_R && fs != tpos - MAIL_R + 1) {
                if (!wait_seq(&out_slot->seq_free, tpos - MAIL_R + 1, h.ctl + 2, i, -1)) wl_fail |= 64;
            }
            if (kept) { out_slot->key[ps] = ckey; out_slot->ts[ps] = cts; out_slot->w[ps] = cw; }
            if (!full && sp >= rsG && sp < drop) out_slot->alt_key[sp - rsG] = ckey;
            const double nn = pre_scale.norm_next;
            if (lane == 0) mail_hdr_write(out_slot, nn, k, uo, mo_, na_, 1);
            const bool fin = mode == FR_RANKS;                   // (then uo == 0: all kept weights distinct)
            if (fin && kept) out_slot->pos[ps] = ps;
            CRIT(2);
            asm volatile("" ::: "memory");
            if (lane == 0) __hip_atomic_store(&out_slot->seq_set, tpos + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (lane == 0 && fin) __hip_atomic_store(&out_slot->seq_ord, tpos + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            CRIT(3);
            __builtin_amdgcn_s_setprio(0);                       // the rest of this hop is off the chain
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
    // the new row in dictionary order is the NEXT position's version: its partner task reads it there
    if (next_by_mail) store_row_scatter_at(hub_version(h, m, chain_idx, tpos + 1), k, lane, n_new, c, new_norm, vtag);
    if (lane == 0)                                               // both stages of the incoming slot have been read
        __hip_atomic_store(&in_slot->seq_free, tpos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
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
__device__ inline void process_chain_partner(const zt_tppr &h, const StreamArgs &A, WaveLds &L, int lane, int i, int mo)
{
    const int k = h.k, B = A.B, n_roles = A.n_roles;
    const int m = A.m_lo + mo;
    const double alpha = h.alpha[m], beta = h.beta[m];
    unsigned *done = h.done + (long long)m * MAX_CHUNK;
    const unsigned *cdone = h.cdone + (long long)m * MAX_CHUNK;
    const unsigned epoch = A.epoch, tag_base = epoch << ORD_BITS, vtag = tag_base | 1u;
    const long long role_stride = A.role_stride;
    const int c = h.owner_of[i], t = h.pos_of[i];
    const long long hub = h.chain_node[c];
    int my_wo = 0, my_pf = -1;
    if (lane < n_roles) { my_wo = h.wo[lane * B + i]; my_pf = h.pflag[lane * B + i]; }
    if (my_pf >= 0) (void)wait_flag(done + my_pf, epoch, h.ctl + 2, my_pf);      // a reader before me has not read yet
    const int wo_u = __shfl(my_wo, 0), wo_v = __shfl(my_wo, 1), wo_g = __shfl(my_wo, 2);
    const long long u = A.nodes[i], v = A.nodes[role_stride + i];
    const long long g = n_roles == 3 ? A.nodes[2 * role_stride + i] : u;
    const double tnow = A.tsv[i];
    const long long e = A.eidx[i];
    const bool hub_is_u = u == hub;
    const long long pnode = hub_is_u ? v : u;          // == hub for a self-loop
    const int wo_p = hub_is_u ? wo_v : wo_u;
    Row rh, rp, rg;
    // ---- rows: the partner's and the negative sample's from memory, the hub's old one from its version slot ----
    const unsigned ptag = wo_p ? (tag_base | (unsigned)wo_p) : 0u, gtag = wo_g ? (tag_base | (unsigned)wo_g) : 0u;
    const bool g_own = n_roles == 3 && g != u && g != v;
    unsigned sp = 0, sg = 0;
    const u64 *ver = hub_version(h, m, c, t);
    unsigned sh = load_row_at(ver, k, lane, vtag, rh);
    if (pnode != hub) sp = load_row(h, m, pnode, lane, ptag, rp);
    if (g_own) sg = load_row(h, m, g, lane, gtag, rg);
    if (pnode != hub && ptag && sp != ptag) (void)load_row_wait(h, m, pnode, lane, ptag, rp, h.ctl + 2);
    if (g_own && gtag && sg != gtag) (void)load_row_wait(h, m, g, lane, gtag, rg, h.ctl + 2);
    {
        unsigned polls = 0;
        long long t0 = 0;
        while (sh != vtag) {                           // the chain has not reached this position yet
            __builtin_amdgcn_s_sleep(32);
            sh = load_row_at(ver, k, lane, vtag, rh);
            if ((++polls & 255u) == 0) {
                const long long now = (long long)wall_clock64();
                if (t0 == 0) t0 = now;
                else if (now - t0 > WAIT_TICKS) { note_timeout(h.ctl + 2, 4, i, (int)vtag, (int)sh, t); break; }
                if (launch_failed(h.ctl + 2)) break;
            }
        }
    }
    // ---- all reads done: later writers of these rows may go ahead ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    st_agent(done + i, epoch);
    if (pnode != hub) {
        // (edge_idx, hub, ts) is the key entering the partner's dictionary
        const u64 nkey = ((u64)(unsigned)e << 32) | (u64)(unsigned)hub;
        Cand cc;
        const int n = merge_pair_reg(L, lane, k, alpha, beta, rp, rh, nkey, tnow, cc);
        const double new_norm = rp.norm * beta + beta;
        (void)wait_flag(cdone + i, epoch, h.ctl + 2, -i - 2);      // the chain has read the partner's old row
        store_row_scatter(h, m, pnode, lane, n, cc, new_norm, tag_base | (unsigned)(wo_p + 1));
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

__global__ __launch_bounds__(WAVE * WAVES_PER_WG) void k_stream(zt_tppr h, StreamArgs A)
{
    __shared__ WaveLds lds[WAVES_PER_WG];
    __shared__ Mail mail;
    __shared__ int ch_edge[CH_MAX], ch_partner[CH_MAX], ch_woh[CH_MAX], ch_wop[CH_MAX], ch_pfh[CH_MAX];   // chain workgroups: HopRec
    WaveLds &L = lds[threadIdx.x / WAVE];
    const int lane = lane_id();
    if (ld_agent(h.ctl + 2) == ZT_ERR_RANGE) {         // rejected by k_count: the state is not touched,
        if (A.emit) {                                  // the output rows read as empty dictionaries
            const int k = h.k;
            for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
                 q < (long long)A.n_models * A.n_roles * A.B * k; q += (long long)gridDim.x * blockDim.x) {
                const long long c = q % k, row = q / k;
                const long long i = row % A.B, role = (row / A.B) % A.n_roles, mo = row / ((long long)A.B * A.n_roles);
                const long long o = (mo * A.out_rows + role * A.role_stride + i) * k + c;
                A.out_nodes[o] = 0; A.out_eidx[o] = 0; A.out_dt[o] = 0.f; A.out_w[o] = 0.f;
            }
        }
        return;
    }
    __builtin_amdgcn_s_setprio(3);                     // chain hops must not queue behind throughput kernels
    if (threadIdx.x < MAIL_R) { mail.slot[threadIdx.x].seq_set = 0; mail.slot[threadIdx.x].seq_ord = 0; mail.slot[threadIdx.x].seq_free = 0; }
    if (threadIdx.x == 0) mail.head = 0;
#ifdef ZT_CRIT
    if (threadIdx.x == 0) mail.t_start = (long long)__builtin_readcyclecounter();
#endif
    for (int q = lane; q < HTAB; q += WAVE) L.htab[q] = -1;
    __syncthreads();
    const int n_models = A.n_models;

    // ---- chain workgroups: blocks [0, chains * models) each own one hub of one model ----
    const int n_chain_wg = A.use_chains ? h.ctl[4] * n_models : 0;
    if ((int)blockIdx.x < n_chain_wg) {
        const int c = blockIdx.x / n_models, mo = blockIdx.x % n_models;
        const long long hub = h.chain_node[c];
        int len = h.chain_len[c];
        len = len < CH_MAX ? len : CH_MAX;
        const int *edges = h.chain_edges + c * CH_MAX;
        // what every hop needs to know about its edge (HopRec), once, into LDS
        for (int q = threadIdx.x; q < len; q += blockDim.x) {
            const int e = edges[q];
            const long long u = A.nodes[e], v = A.nodes[A.role_stride + e];
            const int role_h = u == hub ? 0 : 1;
            ch_edge[q] = e;
            ch_partner[q] = u == v ? -1 : (int)(u == hub ? v : u);
            ch_woh[q] = h.wo[role_h * A.B + e];
            ch_wop[q] = h.wo[(1 - role_h) * A.B + e];
            ch_pfh[q] = h.pflag[role_h * A.B + e];
        }
        __syncthreads();
        // chain_waves (ZT_CHAIN_WAVES, default all eight) waves take hops: what a hop needs besides the hub's update -- the
        // partner's update, the emission -- runs elsewhere (process_chain_partner), but a hop's preparation and its
        // off-chain half (replay, order, stores) still add up to ~5 hop periods of one wave's time.
        if ((int)(threadIdx.x / WAVE) >= A.chain_waves) return;
        ChainHint hint;
        hint.norm_out = 0.0; hint.tpos = -1;
        // (Assigning hop t to wave t mod 8 statically -- so that the SIMD mate of the wave on the chain is the one four
        // hops away -- was measured: the hops then run strictly one after the other, 10x slower.  A wave claiming its next
        // hop right after publishing, to have the partner's row requested early, was measured too: claims then follow
        // the order of publication -- the same fixed rotation -- and a wave with a long off-chain half holds the chain up.)
        for (;;) {
            int t = atomicAdd(&mail.head, lane == 0 ? 1 : 0);     // branch-free (see the general dequeue)
            t = __builtin_amdgcn_readfirstlane(t);
            if (t >= len) break;
            if (t == 0) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(1);   // (the first hop has no mailbox to wait for)
#define ZT_U(x) __builtin_amdgcn_readfirstlane(x)
            const int pe = t > 0 ? ZT_U(ch_edge[t - 1]) : -1, ne = t + 1 < len ? ZT_U(ch_edge[t + 1]) : -1, ce = ZT_U(ch_edge[t]);
            HopRec rec;
            rec.partner = ZT_U(ch_partner[t]); rec.wo_h = ZT_U(ch_woh[t]); rec.wo_p = ZT_U(ch_wop[t]); rec.pf_h = ZT_U(ch_pfh[t]);
            rec.wo_prev = t > 0 ? ZT_U(ch_woh[t - 1]) : -1;
            rec.wo_next = t + 1 < len ? ZT_U(ch_woh[t + 1]) : -1;
            rec.pf_next = t + 1 < len ? ZT_U(ch_pfh[t + 1]) : -1;
#undef ZT_U
            if (!chain_hop(h, A, L, lane, ce, mo, &mail, hub, pe, ne, t, &hint, c, rec))
                process_edge(h, A, L, lane, ce, mo, &mail, hub, pe, ne, t, &hint, c);
        }
        return;                                           // chain workgroups take no general tasks (letting them join the
                                                          // general queue once their chain is done was measured: no difference)
    }

    // ---- general queue: every (edge, model) task not owned by a chain, in order ----
    const int total = A.B * n_models;
    for (;;) {
        // Dequeue with NO divergent branch: every lane issues the add (lane 0
        // adds 1, the rest 0; the compiler folds it into one wave-level atomic).
        // An `if (lane == 0)` here gets jump-threaded with lane-0 code at the
        // end of the previous iteration and the structurizer then replays the
        // body for the remaining lanes (seen in the ISA).
        int idx = atomicAdd(h.ctl + 1, lane == 0 ? 1 : 0);
        idx = __builtin_amdgcn_readfirstlane(idx);
#ifdef ZT_CRIT
        if (idx >= total) {                                // diagnostic: when the last general wave left, and who it was
            if (lane == 0) atomicMax((unsigned long long *)&g_crit[8191 * 16 + 0], (unsigned long long)__builtin_readcyclecounter());
            return;
        }
#else
        if (idx >= total) return;
#endif
        const int i = idx / n_models;
        if (A.use_chains && h.owner_of[i] >= 0) {           // its chain applies the hub's update, this wave the rest
            process_chain_partner(h, A, L, lane, i, idx % n_models);
            continue;
        }
        process_edge(h, A, L, lane, i, idx % n_models, nullptr, -1, -1, -1, 0);
    }
}

// tags -> 0 for every granule (run when the launch epoch wraps)
__global__ void k_retag(u64 *rows, long long n)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        rows[i] &= 0xffffffffull;
}

// point the handle's prepass fields at one of its two sets
void use_set(zt_tppr *h, int q)
{
    const zt_tppr::PlanSet &P = h->set[q];
    h->cnt = P.cnt; h->off = P.off; h->slot = P.slot; h->list = P.list; h->wo = P.wo; h->pflag = P.pflag; h->nxt = P.nxt;
    h->chain_of = P.chain_of; h->hot_node = P.hot_node; h->hot_cnt = P.hot_cnt; h->chain_node = P.chain_node;
    h->chain_len = P.chain_len; h->chain_edges = P.chain_edges; h->owner_of = P.owner_of; h->pos_of = P.pos_of; h->ctl = P.ctl;
}

}  // namespace

// ---- C ABI ------------------------------------------------------------------------
extern "C" int zt_tppr_create(zt_tppr **out, int64_t num_nodes, int32_t k, int32_t n_tppr,
                              const double *alpha_host, const double *beta_host)
{
    if (!out || num_nodes <= 0 || k <= 0 || n_tppr <= 0 || !alpha_host || !beta_host) {
        set_error("zt_tppr_create: bad argument");
        return ZT_ERR_ARG;
    }
    if (k > ZT_MAX_K || n_tppr > 16) {
        set_error("zt_tppr_create: k=%d (max %d) or n_tppr=%d (max 16) unsupported", k, ZT_MAX_K, n_tppr);
        return ZT_ERR_UNSUPPORTED;
    }
    zt_tppr *h = new zt_tppr();
    memset(h, 0, sizeof(*h));
    h->N = num_nodes; h->k = k; h->M = n_tppr; h->rg = HDR + 6 * k;
    for (int m = 0; m < n_tppr; ++m) { h->alpha[m] = alpha_host[m]; h->beta[m] = beta_host[m]; }
    const size_t rows = (size_t)n_tppr * (size_t)num_nodes;
    ZT_HIP(hipMalloc(&h->rows, rows * h->rg * sizeof(u64)));
    ZT_HIP(hipMalloc(&h->done, sizeof(unsigned) * (size_t)MAX_CHUNK * n_tppr));
    ZT_HIP(hipMemset(h->done, 0, sizeof(unsigned) * (size_t)MAX_CHUNK * n_tppr));
    ZT_HIP(hipMalloc(&h->cdone, sizeof(unsigned) * (size_t)MAX_CHUNK * n_tppr));
    ZT_HIP(hipMemset(h->cdone, 0, sizeof(unsigned) * (size_t)MAX_CHUNK * n_tppr));
    h->hubver = nullptr;
    if (k <= REG_K_MAX) {                             // hub chains (register-resident merge) exist for k <= 30 only
        const size_t vb = (size_t)n_tppr * MAX_CHAINS * (CH_MAX + 1) * h->rg * sizeof(u64);
        ZT_HIP(hipMalloc(&h->hubver, vb));
        ZT_HIP(hipMemset(h->hubver, 0, vb));
    }
    for (int q = 0; q < 2; ++q) {
        zt_tppr::PlanSet &P = h->set[q];
        ZT_HIP(hipMalloc(&P.cnt, (size_t)num_nodes * sizeof(int)));
        ZT_HIP(hipMalloc(&P.off, (size_t)num_nodes * sizeof(int)));
        ZT_HIP(hipMalloc(&P.ctl, CTL_WORDS * sizeof(int)));
        ZT_HIP(hipMalloc(&P.slot, sizeof(int) * 3 * MAX_CHUNK));
        ZT_HIP(hipMalloc(&P.list, sizeof(int) * 3 * MAX_CHUNK));
        ZT_HIP(hipMalloc(&P.wo, sizeof(int) * 3 * MAX_CHUNK));
        ZT_HIP(hipMalloc(&P.pflag, sizeof(int) * 3 * MAX_CHUNK));
        ZT_HIP(hipMalloc(&P.nxt, sizeof(int) * 3 * MAX_CHUNK));
        ZT_HIP(hipMalloc(&P.chain_of, (size_t)num_nodes * sizeof(int)));
        ZT_HIP(hipMalloc(&P.hot_node, sizeof(int) * MAX_HOT));
        ZT_HIP(hipMalloc(&P.hot_cnt, sizeof(int) * MAX_HOT));
        ZT_HIP(hipMalloc(&P.chain_node, sizeof(int) * MAX_CHAINS));
        ZT_HIP(hipMalloc(&P.chain_len, sizeof(int) * MAX_CHAINS));
        ZT_HIP(hipMalloc(&P.chain_edges, sizeof(int) * MAX_CHAINS * CH_MAX));
        ZT_HIP(hipMalloc(&P.owner_of, sizeof(int) * MAX_CHUNK));
        ZT_HIP(hipMalloc(&P.pos_of, sizeof(int) * MAX_CHUNK));
        ZT_HIP(hipMemset(P.chain_of, 0xff, (size_t)num_nodes * sizeof(int)));
        ZT_HIP(hipMemset(P.chain_len, 0, sizeof(int) * MAX_CHAINS));
        ZT_HIP(hipMemset(P.cnt, 0, (size_t)num_nodes * sizeof(int)));
        ZT_HIP(hipMemset(P.ctl, 0, CTL_WORDS * sizeof(int)));
        ZT_HIP(hipEventCreateWithFlags(&P.planned, hipEventDisableTiming | zt::sync_event_flags()));
        ZT_HIP(hipEventCreateWithFlags(&P.consumed, hipEventDisableTiming | zt::sync_event_flags()));
        P.used = false;
        P.valid = false;
    }
    ZT_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->latch_host), sizeof(int), hipHostMallocMapped));
    *h->latch_host = 0;
    ZT_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&h->latch_dev), h->latch_host, 0));
    for (int q = 0; q < 2; ++q)                        // note_timeout finds the latch through ctl[14..15]
        ZT_HIP(hipMemcpy(h->set[q].ctl + 14, &h->latch_dev, sizeof(int *), hipMemcpyHostToDevice));
    h->plan_serial = 0;
    h->next_set = 0;
    use_set(h, 0);
    hipDeviceProp_t prop;
    int dev = 0;
    ZT_HIP(hipGetDevice(&dev));
    ZT_HIP(hipGetDeviceProperties(&prop, dev));
    h->n_cu = prop.multiProcessorCount;
    // every workgroup of a k_stream grid that runs hub chains must be resident: ask the runtime how many
    // fit on a CU (LDS, registers) rather than estimating it
    int per_cu = 0;
    ZT_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_stream, WAVE * WAVES_PER_WG, 0));
    h->wg_per_cu = per_cu;
    h->epoch = 0;
    int rc = zt_tppr_reset(h, nullptr);
    if (rc != ZT_OK) return rc;
    ZT_HIP(hipDeviceSynchronize());
    *out = h;
    return ZT_OK;
}

extern "C" int zt_tppr_destroy(zt_tppr *h)
{
    if (!h) return ZT_OK;
    (void)hipDeviceSynchronize();
    (void)hipFree(h->rows); (void)hipFree(h->done); (void)hipFree(h->cdone);
    if (h->hubver) (void)hipFree(h->hubver);
    if (h->latch_host) (void)hipHostFree(h->latch_host);
    for (int q = 0; q < 2; ++q) {
        zt_tppr::PlanSet &P = h->set[q];
        (void)hipFree(P.cnt); (void)hipFree(P.off); (void)hipFree(P.ctl); (void)hipFree(P.slot); (void)hipFree(P.list);
        (void)hipFree(P.wo); (void)hipFree(P.pflag); (void)hipFree(P.nxt); (void)hipFree(P.chain_of);
        (void)hipFree(P.hot_node); (void)hipFree(P.hot_cnt); (void)hipFree(P.chain_node); (void)hipFree(P.chain_len);
        (void)hipFree(P.chain_edges); (void)hipFree(P.owner_of); (void)hipFree(P.pos_of);
        (void)hipEventDestroy(P.planned); (void)hipEventDestroy(P.consumed);
    }
    delete h;
    return ZT_OK;
}

extern "C" int zt_tppr_reset(zt_tppr *h, void *stream)
{
    if (!h) return ZT_ERR_ARG;
    const size_t rows = (size_t)h->M * (size_t)h->N;
    ZT_HIP(hipMemsetAsync(h->rows, 0, rows * h->rg * sizeof(u64), (hipStream_t)stream));
    h->set[0].valid = h->set[1].valid = false;       // plans made for the old state's stream are void
    return ZT_OK;
}

extern "C" int zt_tppr_copy(zt_tppr *dst, const zt_tppr *src, void *stream)
{
    if (!dst || !src) return ZT_ERR_ARG;
    if (dst->N != src->N || dst->k != src->k || dst->M != src->M) {
        set_error("zt_tppr_copy: shape mismatch");
        return ZT_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)src->M * (size_t)src->N * src->rg;
    ZT_HIP(hipMemcpyAsync(dst->rows, src->rows, n * sizeof(u64), hipMemcpyDeviceToDevice, s));
    // tags are relative to the owner's launch epoch: strip them in the copy
    k_retag<<<2048, 256, 0, s>>>(dst->rows, (long long)n);
    ZT_LAUNCH_CHECK();
    dst->set[0].valid = dst->set[1].valid = false;
    return ZT_OK;
}

// CUs a stream may use (CU-masked streams: the size of the mask)
static int stream_cus(const zt_tppr *h, hipStream_t s)
{
    uint32_t mask[32] = {0};
    if (hipExtStreamGetCUMask(s, 32, mask) != hipSuccess) { (void)hipGetLastError(); return h->n_cu; }
    int c = 0;
    for (int q = 0; q < 32; ++q) c += __builtin_popcount(mask[q]);
    return (c > 0 && c < h->n_cu) ? c : h->n_cu;
}

// hub chains a grid can carry: at most two thirds of its workgroups, so the general queue always keeps waves
static int chains_for_grid(int grid, int n_models)
{
    static const int chains_env = getenv("ZT_STREAM_CHAINS") ? atoi(getenv("ZT_STREAM_CHAINS")) : MAX_CHAINS;
    int max_chains = (2 * grid) / (3 * n_models);
    if (max_chains > chains_env) max_chains = chains_env;
    if (max_chains > MAX_CHAINS) max_chains = MAX_CHAINS;
    return max_chains < 0 ? 0 : max_chains;
}

// grid of k_stream and the number of hub chains for a launch of B edges on a stream of `cus` CUs
static void launch_shape(const zt_tppr *h, int cus, int B, int n_models, int *grid_out, int *max_chains_out)
{
    long long waves = (long long)B * n_models;
    // The kernel is latency-bound (waves mostly sleep on their predecessors): a few waves per CU
    // drain the independent tasks fast enough, and leave LDS / issue slots to a concurrently
    // running aggregation kernel.  ZT_STREAM_WGS_PER_CU overrides (workgroups of 8 waves).
    static const double wgs_per_cu = getenv("ZT_STREAM_WGS_PER_CU") ? atof(getenv("ZT_STREAM_WGS_PER_CU")) : 1.0;
    // Every workgroup of the grid must be resident at once (chain workgroups wait on each other's rows):
    // the CUs of the stream that runs k_stream (a CU-masked stream offers fewer) times the workgroups one
    // CU holds (asked from the runtime at create time).
    long long max_waves = (long long)(cus * WAVES_PER_WG * wgs_per_cu);
    const long long resident = (long long)cus * (h->wg_per_cu > 0 ? h->wg_per_cu : 1) * WAVES_PER_WG;
    if (max_waves > resident) max_waves = resident;
    if (max_waves < WAVES_PER_WG) max_waves = WAVES_PER_WG;
    if (waves > max_waves) waves = max_waves;
    const int grid = (int)((waves + WAVES_PER_WG - 1) / WAVES_PER_WG);
    *grid_out = grid;
    // the chain hand-off (two-stage mailbox) is built on the register-resident merge: k <= 30
    *max_chains_out = h->k <= REG_K_MAX ? chains_for_grid(grid, n_models) : 0;
}

// The dependency prepass of one launch into plan set q, on stream s.  It reads only the node and
// edge ids, never the T-PPR rows, so it may run while k_stream works on the other set.
static int plan_chunk(zt_tppr *h, int q, const int32_t *nodes, const long long *eidx, long long role_stride, int B,
                      int n_roles, int model, hipStream_t s)
{
    zt_tppr::PlanSet &P = h->set[q];
    if (P.used) ZT_HIP(hipStreamWaitEvent(s, P.consumed, 0));      // k_stream of two calls ago has let go of it
    use_set(h, q);
    const int A = B * n_roles;
    const int tb = 256, gb = (A + tb - 1) / tb;
    const int n_models = model < 0 ? h->M : 1;
    int grid, max_chains;
    launch_shape(h, h->run_cus > 0 ? h->run_cus : h->n_cu, B, n_models, &grid, &max_chains);
    static const bool fused_ok = !(getenv("ZT_PREPASS_FUSED") && atoi(getenv("ZT_PREPASS_FUSED")) == 0);
    if (fused_ok && A <= PRE_FUSED_MAX) {
        ZT_PROF_BEGIN(s, P_PREPASS);
        k_prepass_fused<<<1, PRE_THREADS, 0, s>>>(nodes, eidx, role_stride, B, n_roles, h->N, h->cnt, h->slot, h->off, h->list,
                                                  h->wo, h->pflag, h->nxt, h->ctl, h->latch_dev, h->hot_node, h->hot_cnt,
                                                  h->chain_of, h->chain_node, h->chain_len, h->chain_edges, h->owner_of,
                                                  h->pos_of, max_chains);
        ZT_PROF_END(s, P_PREPASS);
    } else {
        ZT_PROF_BEGIN(s, P_PREPASS);
        k_plan_begin<<<1, 64, 0, s>>>(h->ctl);
        k_count<<<gb, tb, 0, s>>>(nodes, eidx, role_stride, B, n_roles, h->N, h->cnt, h->slot, h->ctl, h->latch_dev);
        k_reserve<<<gb, tb, 0, s>>>(nodes, role_stride, B, n_roles, h->cnt, h->off, h->slot, h->ctl, h->hot_node, h->hot_cnt);
        k_fill<<<gb, tb, 0, s>>>(nodes, role_stride, B, n_roles, h->off, h->slot, h->list);
        k_deps<<<gb, tb, 0, s>>>(nodes, role_stride, B, n_roles, h->cnt, h->off, h->slot, h->list, h->wo, h->pflag, h->nxt);
        k_hot_select<<<1, MAX_HOT, 0, s>>>(h->ctl, h->hot_node, h->hot_cnt, h->chain_of, h->chain_node, h->chain_len,
                                           max_chains);
        k_own<<<(B + tb - 1) / tb, tb, 0, s>>>(nodes, role_stride, B, h->cnt, h->slot, h->chain_of, h->chain_len,
                                               h->chain_edges, h->owner_of);
        if (max_chains > 0) k_chain_sort<<<max_chains, 64, 0, s>>>(h->ctl, h->chain_len, h->chain_edges, h->pos_of);
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

// k_stream over plan set q (planned on any stream), on stream s.
static int run_chunk(zt_tppr *h, int q, const int32_t *nodes, const double *ts, const long long *eidx,
                     long long role_stride, int B, int n_roles, int emit, int model, long long out_rows, int32_t *on,
                     int32_t *oe, float *od, float *ow, hipStream_t s, bool plan_ordered = false,
                     hipEvent_t *done_out = nullptr, int sub_B = 0)
{
    zt_tppr::PlanSet &P = h->set[q];
    if (h->epoch >= EPOCH_MAX) {               // launch epoch about to wrap: forget all tags
        k_retag<<<2048, 256, 0, s>>>(h->rows, (long long)h->M * h->N * h->rg);
        ZT_HIP(hipMemsetAsync(h->done, 0, sizeof(unsigned) * (size_t)MAX_CHUNK * h->M, s));   // flags of old epochs
        ZT_HIP(hipMemsetAsync(h->cdone, 0, sizeof(unsigned) * (size_t)MAX_CHUNK * h->M, s));
        if (h->hubver) ZT_HIP(hipMemsetAsync(h->hubver, 0, (size_t)h->M * MAX_CHAINS * (CH_MAX + 1) * h->rg * sizeof(u64), s));
        h->epoch = 0;
    }
    h->epoch += 1;
    if (!plan_ordered) ZT_HIP(hipStreamWaitEvent(s, P.planned, 0));      // (the caller has ordered s behind the plan already)
    use_set(h, q);
    h->run_cus = stream_cus(h, s);
    // The plan sized the grid for the CUs it expected.  If THIS stream offers fewer (a CU mask the plan did
    // not know about), shrink the grid to what is resident here; hub chains only run when their workgroups
    // plus a general queue fit, otherwise every edge goes through the in-order queue, which needs no
    // residency (a task only waits on tasks already dequeued by resident waves).
    int grid = P.grid, use_chains = 1;
    const int resident = h->run_cus * (h->wg_per_cu > 0 ? h->wg_per_cu : 1);
    if (grid > resident) {
        grid = resident;
        use_chains = P.max_chains <= chains_for_grid(grid, model < 0 ? h->M : 1) ? 1 : 0;
    }
    StreamArgs sa;
    sa.use_chains = use_chains;
    sa.nodes = nodes; sa.tsv = ts; sa.eidx = eidx; sa.role_stride = role_stride; sa.B = B; sa.n_roles = n_roles;
    sa.emit = emit; sa.m_lo = model < 0 ? 0 : model; sa.n_models = model < 0 ? h->M : 1; sa.out_rows = out_rows;
    sa.out_nodes = on; sa.out_eidx = oe; sa.out_dt = od; sa.out_w = ow; sa.epoch = h->epoch;
    sa.sub_B = sub_B;
    static const int chain_waves_env = getenv("ZT_CHAIN_WAVES") ? atoi(getenv("ZT_CHAIN_WAVES")) : 8;
    sa.chain_waves = chain_waves_env < 1 ? 1 : (chain_waves_env > WAVES_PER_WG ? WAVES_PER_WG : chain_waves_env);
    static const int crit_multi_env = getenv("ZT_CRIT_MULTI") ? atoi(getenv("ZT_CRIT_MULTI")) : 0;
    sa.crit_multi = crit_multi_env;
#ifdef ZT_WAITLOG
    {
        void *wl = nullptr;
        ZT_HIP(hipGetSymbolAddress(&wl, HIP_SYMBOL(g_wl)));
        ZT_HIP(hipMemsetAsync(wl, 0, sizeof(int) * 2 * MAX_CHUNK * 8, s));
        h->dbg_nodes = nodes; h->dbg_stride = role_stride; h->dbg_B = B; h->dbg_roles = n_roles; h->dbg_models = sa.n_models;
    }
#endif
    ZT_PROF_BEGIN(s, P_STREAM);
    k_stream<<<grid, WAVE * WAVES_PER_WG, 0, s>>>(*h, sa);
    ZT_PROF_END(s, P_STREAM);
    ZT_LAUNCH_CHECK();
    ZT_HIP(hipEventRecord(P.consumed, s));
    if (done_out) *done_out = P.consumed;
    P.used = true;
    P.valid = false;
    return ZT_OK;
}

// a failure latched by an earlier launch makes every later call fail until zt_tppr_status has reported it
static int latched(const zt_tppr *h, const char *who)
{
    const int st = *reinterpret_cast<volatile int *>(h->latch_host);
    if (st == 0) return ZT_OK;
    set_error("%s: an earlier launch on this handle failed (%s); zt_tppr_status reports and clears it", who,
              st == ZT_ERR_RANGE ? "node or edge id out of range, that batch was not applied" : "dependency wait timed out");
    return st;
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
    if (!nodes_dev || !eidx_dev) { set_error("zt_tppr_plan: NULL buffer"); return ZT_ERR_ARG; }
    const int q = h->next_set;
    h->next_set ^= 1;
    h->set[q].valid = false;
    int rc = plan_chunk(h, q, nodes_dev, reinterpret_cast<const long long *>(eidx_dev), B, (int)B, n_roles, model,
                        (hipStream_t)stream);
    if (rc != ZT_OK) return rc;
    h->set[q].valid = true;
    h->set[q].token = ++h->plan_serial;
    *token_out = h->set[q].token;
    return ZT_OK;
}

// zt_tppr_stream with two extras for callers inside the library (pipeline.hip): plan_ordered = `stream` already
// waits for the stream that made the plan (no second wait packet); *done_out = the event recorded behind the
// (last) update kernel, so that the caller need not record one of its own.  Every packet between two update
// kernels on the T-PPR stream costs ~5 us of the step.
int zt::tppr_stream_ex(zt_tppr *h, const int32_t *nodes_dev, const double *ts_dev, const int64_t *eidx_dev, int64_t B,
                       int32_t n_roles, int32_t emit, int32_t model, int32_t *out_nodes_dev, int32_t *out_eidx_dev,
                       float *out_dt_dev, float *out_w_dev, uint64_t plan_token, void *stream, bool plan_ordered,
                       hipEvent_t *done_out, int32_t sub_B)
{
    if (done_out) *done_out = nullptr;
    if (sub_B < 0 || (sub_B > 0 && (B > MAX_CHUNK || sub_B > B))) {
        set_error("zt_tppr_stream: a launch over several batches must fit one chunk (%d edges)", MAX_CHUNK);
        return ZT_ERR_ARG;
    }
    if (!h || B < 0 || (n_roles != 2 && n_roles != 3) || model >= h->M) {
        set_error("zt_tppr_stream: bad argument");
        return ZT_ERR_ARG;
    }
    if (int st = latched(h, "zt_tppr_stream")) return st;
    if (B == 0) return ZT_OK;
    if (!nodes_dev || !ts_dev || !eidx_dev ||
        (emit && (!out_nodes_dev || !out_eidx_dev || !out_dt_dev || !out_w_dev))) {
        set_error("zt_tppr_stream: NULL buffer");
        return ZT_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    const long long *e64 = reinterpret_cast<const long long *>(eidx_dev);
    // a prepass made ahead of time by zt_tppr_plan for exactly this call: the caller hands back the token
    // that plan returned (a recycled device address alone must never select a stale plan)
    if (plan_token != 0) {
        for (int q = 0; q < 2; ++q) {
            zt_tppr::PlanSet &P = h->set[q];
            if (P.valid && P.token == plan_token) {
                if (P.nodes != nodes_dev || P.B != (int)B || P.n_roles != n_roles || P.model != model) {
                    set_error("zt_tppr_stream: plan token %llu was made for another call", (unsigned long long)plan_token);
                    return ZT_ERR_ARG;
                }
                return run_chunk(h, q, nodes_dev, ts_dev, e64, B, (int)B, n_roles, emit, model, (long long)n_roles * B,
                                 out_nodes_dev, out_eidx_dev, out_dt_dev, out_w_dev, s, plan_ordered, done_out, sub_B);
            }
        }
        // the plan is gone (reset / copy / import, or two newer plans): fall through to an inline prepass
    }
    h->run_cus = stream_cus(h, s);             // plan and run on the same stream here
    // launches of at most MAX_CHUNK edges: writer ordinals must fit the tag
    for (int64_t c0 = 0; c0 < B; c0 += MAX_CHUNK) {
        const int bc = (int)((B - c0) < MAX_CHUNK ? (B - c0) : MAX_CHUNK);
        const size_t oo = (size_t)c0 * h->k;
        const int q = h->next_set;
        h->next_set ^= 1;
        h->set[q].valid = false;
        int rc = plan_chunk(h, q, nodes_dev + c0, e64 + c0, B, bc, n_roles, model, s);
        if (rc != ZT_OK) return rc;
        rc = run_chunk(h, q, nodes_dev + c0, ts_dev + c0, e64 + c0, B, bc, n_roles, emit, model,
                       (long long)n_roles * B, emit ? out_nodes_dev + oo : nullptr, emit ? out_eidx_dev + oo : nullptr,
                       emit ? out_dt_dev + oo : nullptr, emit ? out_w_dev + oo : nullptr, s, true, done_out, sub_B);   // planned on s itself
        if (rc != ZT_OK) return rc;
    }
    return ZT_OK;
}

extern "C" int zt_tppr_stream(zt_tppr *h, const int32_t *nodes_dev, const double *ts_dev, const int64_t *eidx_dev,
                              int64_t B, int32_t n_roles, int32_t emit, int32_t model, int32_t *out_nodes_dev,
                              int32_t *out_eidx_dev, float *out_dt_dev, float *out_w_dev, uint64_t plan_token,
                              void *stream)
{
    return zt::tppr_stream_ex(h, nodes_dev, ts_dev, eidx_dev, B, n_roles, emit, model, out_nodes_dev, out_eidx_dev,
                              out_dt_dev, out_w_dev, plan_token, stream, false, nullptr, 0);
}

#ifdef ZT_CRIT
extern "C" int zt_debug_crit(long long *host, int n)
{
    ZT_HIP(hipDeviceSynchronize());
    ZT_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_crit), sizeof(long long) * n * 16));
    return ZT_OK;
}
#endif
#ifdef ZT_STAMP
extern "C" int zt_debug_stamps(long long *host, int n)
{
    ZT_HIP(hipDeviceSynchronize());
    ZT_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), sizeof(long long) * n * 4));
    return ZT_OK;
}
extern "C" int zt_debug_paths(int *host)
{
    ZT_HIP(hipDeviceSynchronize());
    ZT_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_paths), sizeof(int) * 8));
    return ZT_OK;
}
extern "C" int zt_debug_stamps2(long long *host, int n)
{
    ZT_HIP(hipDeviceSynchronize());
    ZT_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps2), sizeof(long long) * n * 8));
    return ZT_OK;
}
#endif

// A HIP stream restricted to the compute units [cu_lo, cu_hi): lets the latency-bound T-PPR
// kernel run beside the throughput-bound aggregation without sharing CUs with it.
extern "C" int zt_stream_create_masked(void **out, int32_t cu_lo, int32_t cu_hi)
{
    if (!out || cu_lo < 0 || cu_hi <= cu_lo) { set_error("zt_stream_create_masked: bad argument"); return ZT_ERR_ARG; }
    hipDeviceProp_t prop;
    int dev = 0;
    ZT_HIP(hipGetDevice(&dev));
    ZT_HIP(hipGetDeviceProperties(&prop, dev));
    const int n = prop.multiProcessorCount;
    if (cu_hi > n) cu_hi = n;
    std::vector<uint32_t> mask((n + 31) / 32, 0u);
    for (int c = cu_lo; c < cu_hi; ++c) mask[c / 32] |= 1u << (c % 32);
    hipStream_t s;
    ZT_HIP(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
    *out = s;
    return ZT_OK;
}

extern "C" int zt_stream_destroy(void *stream)
{
    if (stream) ZT_HIP(hipStreamDestroy((hipStream_t)stream));
    return ZT_OK;
}

// test hook: move the launch epoch (e.g. next to its wrap-around)
extern "C" int zt_test_set_epoch(zt_tppr *h, uint32_t epoch)
{
    if (!h) return ZT_ERR_ARG;
    h->epoch = epoch > EPOCH_MAX ? EPOCH_MAX : epoch;
    return ZT_OK;
}

extern "C" int zt_tppr_status(zt_tppr *h, void *stream)
{
    if (!h) return ZT_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    int c[CTL_WORDS] = {0};
    int st = 0;
    ZT_HIP(hipStreamSynchronize(s));
    const int latch = *reinterpret_cast<volatile int *>(h->latch_host);
    *reinterpret_cast<volatile int *>(h->latch_host) = 0;
    for (int q = 0; q < 2 && st == 0; ++q) {          // the status word of either plan set (details of a time-out)
        ZT_HIP(hipMemcpyAsync(c, h->set[q].ctl, sizeof(c), hipMemcpyDeviceToHost, s));
        ZT_HIP(hipStreamSynchronize(s));
        st = c[2];
        if (st != 0) {
            ZT_HIP(hipMemsetAsync(h->set[q].ctl + 2, 0, sizeof(int), s));
            ZT_HIP(hipMemsetAsync(h->set[q].ctl + 13, 0, sizeof(int), s));
            ZT_HIP(hipStreamSynchronize(s));
            use_set(h, q);
        }
    }
    if (st == 0 && latch != 0) {                      // the failing set has been re-planned since: the latch remembers
        st = latch;
        c[13] = 0;
    }
    if (st != 0) {
        if (st == ZT_ERR_RANGE) {
            set_error("node or edge id out of range");
        } else {
            // kind 1: reads-done flag, 2: row tag (node, expect, seen, model), 3: mailbox (edge, expect, seen, prev edge)
            char msg[480];
            int o = snprintf(msg, sizeof(msg), "dependency wait timed out; %d waits gave up:", c[13]);
            for (int q = 0; q < CTL_LOG && q < c[13] && o < (int)sizeof(msg) - 80; ++q) {
                const int *r = c + 16 + 8 * q;
                o += snprintf(msg + o, sizeof(msg) - o, " [kind %d: %d expect 0x%x seen 0x%x aux %d wg %d wave %d t %d]", r[0],
                              r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
            }
            set_error("%s", msg);
#ifdef ZT_WAITLOG
            {
                const int B = h->dbg_B, R = h->dbg_roles;
                std::vector<int> wl(2 * MAX_CHUNK * 8), nodes((size_t)R * B), wo(3 * MAX_CHUNK), own(MAX_CHUNK), cn(MAX_CHAINS),
                    cl(MAX_CHAINS), ce(MAX_CHAINS * CH_MAX);
                (void)hipMemcpyFromSymbol(wl.data(), HIP_SYMBOL(g_wl), wl.size() * 4);
                for (int r = 0; r < R; ++r)
                    (void)hipMemcpy(nodes.data() + (size_t)r * B, h->dbg_nodes + r * h->dbg_stride, (size_t)B * 4, hipMemcpyDeviceToHost);
                (void)hipMemcpy(wo.data(), h->wo, wo.size() * 4, hipMemcpyDeviceToHost);
                (void)hipMemcpy(own.data(), h->owner_of, own.size() * 4, hipMemcpyDeviceToHost);
                (void)hipMemcpy(cn.data(), h->chain_node, cn.size() * 4, hipMemcpyDeviceToHost);
                (void)hipMemcpy(cl.data(), h->chain_len, cl.size() * 4, hipMemcpyDeviceToHost);
                (void)hipMemcpy(ce.data(), h->chain_edges, ce.size() * 4, hipMemcpyDeviceToHost);
                fprintf(stderr, "[waitlog] B %d roles %d models %d ctl4(after reset) %d\n", B, R, h->dbg_models, c[4]);
                for (int q = 0; q < MAX_CHAINS; ++q) {
                    fprintf(stderr, "[waitlog] chain %d node %d len %d:", q, cn[q], cl[q]);
                    for (int t = 0; t < cl[q] && t < 12; ++t) fprintf(stderr, " %d", ce[q * CH_MAX + t]);
                    fprintf(stderr, "\n");
                }
                for (int mo = 0; mo < h->dbg_models && mo < 2; ++mo) {
                    int tmin = 0x7fffffff;
                    for (int i = 0; i < B; ++i) {
                        const int *w = wl.data() + (mo * MAX_CHUNK + i) * 8;
                        if (w[0] && w[2] < tmin) tmin = w[2];
                    }
                    int shown = 0;
                    for (int i = 0; i < B && shown < 40; ++i) {
                        const int *w = wl.data() + (mo * MAX_CHUNK + i) * 8;
                        const bool slow = w[0] == 9 && (w[4] - w[2]) > 20000;     // > 25 ms
                        if (w[0] == 9 && !slow && (w[6] & 255) == 0) continue;
                        ++shown;
                        fprintf(stderr, "[waitlog] m%d edge %d (u %d v %d wo %d %d owner %d): state %d wg %d wave %d start %d rows +%d end +%d prev %d fail 0x%x seen 0x%x\n",
                                mo, i, nodes[i], nodes[B + i], wo[i], wo[B + i], own[i], w[0], w[1] / WAVES_PER_WG, w[1] % WAVES_PER_WG, w[2] - tmin,
                                w[3] - w[2], w[4] - w[2], w[5], w[6], w[7]);
                    }
                }
            }
#endif
        }
    }
    return st;
}

namespace {
// granules of one row -> the reference's dictionary items in iteration order (zeros beyond len)
void decode_row(const u64 *r, size_t k, int32_t *len_out, double *norm_out, int64_t *eidx, int64_t *node, double *ts,
                double *w)
{
    auto lo = [](u64 x) { return (u64)(unsigned)x; };
    *len_out = (int)(unsigned)r[0];
    const u64 nb = lo(r[1]) | (lo(r[2]) << 32);
    memcpy(norm_out, &nb, 8);
    for (size_t j = 0; j < k; ++j) {
        const u64 *e = r + HDR + j;
        if ((int)j < *len_out) {
            node[j] = (int64_t)lo(e[0]);
            eidx[j] = (int64_t)lo(e[k]);
            const u64 tb = lo(e[2 * k]) | (lo(e[3 * k]) << 32), wb = lo(e[4 * k]) | (lo(e[5 * k]) << 32);
            memcpy(&ts[j], &tb, 8);
            memcpy(&w[j], &wb, 8);
        } else {
            eidx[j] = 0; node[j] = 0; ts[j] = 0.0; w[j] = 0.0;
        }
    }
}

__global__ void k_gather_rows(const u64 *__restrict__ rows, const long long *__restrict__ ids, long long n, int rg,
                              u64 *__restrict__ out)
{
    const long long r = blockIdx.x;
    if (r >= n) return;
    const u64 *src = rows + ids[r] * rg;
    for (int c = threadIdx.x; c < rg; c += blockDim.x) out[r * rg + c] = src[c];
}
}  // namespace

extern "C" int zt_tppr_export(zt_tppr *h, int32_t m, int32_t *len_host, double *norm_host, int64_t *eidx_host,
                              int64_t *node_host, double *ts_host, double *w_host)
{
    if (!h || m < 0 || m >= h->M) return ZT_ERR_ARG;
    ZT_HIP(hipDeviceSynchronize());
    const size_t N = (size_t)h->N, k = (size_t)h->k, rg = (size_t)h->rg;
    std::vector<u64> g(N * rg);
    ZT_HIP(hipMemcpy(g.data(), h->rows + (size_t)m * N * rg, N * rg * sizeof(u64), hipMemcpyDeviceToHost));
    for (size_t v = 0; v < N; ++v)
        decode_row(g.data() + v * rg, k, &len_host[v], &norm_host[v], eidx_host + v * k, node_host + v * k,
                   ts_host + v * k, w_host + v * k);
    return ZT_OK;
}

extern "C" int zt_tppr_export_rows(zt_tppr *h, int32_t m, const int64_t *ids_host, int64_t n, int32_t *len_host,
                                   double *norm_host, int64_t *eidx_host, int64_t *node_host, double *ts_host,
                                   double *w_host)
{
    if (!h || m < 0 || m >= h->M || n < 0 || (n > 0 && !ids_host)) return ZT_ERR_ARG;
    if (n == 0) return ZT_OK;
    for (int64_t q = 0; q < n; ++q)
        if (ids_host[q] < 0 || ids_host[q] >= h->N) { set_error("zt_tppr_export_rows: id out of range"); return ZT_ERR_RANGE; }
    ZT_HIP(hipDeviceSynchronize());
    const size_t k = (size_t)h->k, rg = (size_t)h->rg;
    long long *ids_dev = nullptr;
    u64 *buf_dev = nullptr;
    ZT_HIP(hipMalloc(&ids_dev, (size_t)n * sizeof(long long)));
    ZT_HIP(hipMalloc(&buf_dev, (size_t)n * rg * sizeof(u64)));
    ZT_HIP(hipMemcpy(ids_dev, ids_host, (size_t)n * sizeof(long long), hipMemcpyHostToDevice));
    k_gather_rows<<<(unsigned)n, 128>>>(h->rows + (size_t)m * (size_t)h->N * rg, ids_dev, n, (int)rg, buf_dev);
    std::vector<u64> g((size_t)n * rg);
    hipError_t e = hipMemcpy(g.data(), buf_dev, (size_t)n * rg * sizeof(u64), hipMemcpyDeviceToHost);
    (void)hipFree(ids_dev); (void)hipFree(buf_dev);
    ZT_HIP(e);
    for (size_t v = 0; v < (size_t)n; ++v)
        decode_row(g.data() + v * rg, k, &len_host[v], &norm_host[v], eidx_host + v * k, node_host + v * k,
                   ts_host + v * k, w_host + v * k);
    return ZT_OK;
}

namespace {
// the reference's dictionary items (iteration order) -> granules of one row, tag 0
int encode_row(const zt_tppr *h, u64 *r, int32_t len, double norm, const int64_t *eidx, const int64_t *node,
               const double *ts, const double *w)
{
    const size_t k = (size_t)h->k;
    if (len < 0 || len > (int)k) { set_error("zt_tppr_import: bad length"); return ZT_ERR_ARG; }
    u64 nb;
    memcpy(&nb, &norm, 8);
    r[0] = (u64)(unsigned)len; r[1] = (u64)(unsigned)nb; r[2] = nb >> 32;
    for (size_t j = 0; j < (size_t)len; ++j) {
        if (eidx[j] < 0 || eidx[j] > 0x7fffffffll || node[j] < 0 || node[j] >= h->N) {
            set_error("zt_tppr_import: id out of range");
            return ZT_ERR_RANGE;
        }
        u64 tb, wb;
        memcpy(&tb, &ts[j], 8);
        memcpy(&wb, &w[j], 8);
        u64 *e = r + HDR + j;
        e[0] = (u64)node[j]; e[k] = (u64)eidx[j];
        e[2 * k] = (u64)(unsigned)tb; e[3 * k] = tb >> 32;
        e[4 * k] = (u64)(unsigned)wb; e[5 * k] = wb >> 32;
    }
    return ZT_OK;
}

__global__ void k_scatter_tppr_rows(u64 *__restrict__ rows, const long long *__restrict__ ids, long long n, int rg,
                                    const u64 *__restrict__ in)
{
    const long long r = blockIdx.x;
    if (r >= n) return;
    u64 *dst = rows + ids[r] * rg;
    for (int c = threadIdx.x; c < rg; c += blockDim.x) dst[c] = in[r * rg + c];
}
}  // namespace

extern "C" int zt_tppr_import(zt_tppr *h, int32_t m, const int32_t *len_host, const double *norm_host,
                              const int64_t *eidx_host, const int64_t *node_host, const double *ts_host,
                              const double *w_host)
{
    if (!h || m < 0 || m >= h->M) return ZT_ERR_ARG;
    const size_t N = (size_t)h->N, k = (size_t)h->k, rg = (size_t)h->rg;
    std::vector<u64> g(N * rg, 0ull);
    for (size_t v = 0; v < N; ++v) {
        int rc = encode_row(h, g.data() + v * rg, len_host[v], norm_host[v], eidx_host + v * k, node_host + v * k,
                            ts_host + v * k, w_host + v * k);
        if (rc != ZT_OK) return rc;
    }
    ZT_HIP(hipDeviceSynchronize());
    ZT_HIP(hipMemcpy(h->rows + (size_t)m * N * rg, g.data(), N * rg * sizeof(u64), hipMemcpyHostToDevice));
    h->set[0].valid = h->set[1].valid = false;
    return ZT_OK;
}

extern "C" int zt_tppr_import_rows(zt_tppr *h, int32_t m, const int64_t *ids_host, int64_t n, const int32_t *len_host,
                                   const double *norm_host, const int64_t *eidx_host, const int64_t *node_host,
                                   const double *ts_host, const double *w_host)
{
    if (!h || m < 0 || m >= h->M || n < 0 || (n > 0 && !ids_host)) return ZT_ERR_ARG;
    if (n == 0) return ZT_OK;
    const size_t k = (size_t)h->k, rg = (size_t)h->rg;
    std::vector<u64> g((size_t)n * rg, 0ull);
    for (size_t v = 0; v < (size_t)n; ++v) {
        if (ids_host[v] < 0 || ids_host[v] >= h->N) { set_error("zt_tppr_import_rows: id out of range"); return ZT_ERR_RANGE; }
        int rc = encode_row(h, g.data() + v * rg, len_host[v], norm_host[v], eidx_host + v * k, node_host + v * k,
                            ts_host + v * k, w_host + v * k);
        if (rc != ZT_OK) return rc;
    }
    ZT_HIP(hipDeviceSynchronize());
    long long *ids_dev = nullptr;
    u64 *buf_dev = nullptr;
    ZT_HIP(hipMalloc(&ids_dev, (size_t)n * sizeof(long long)));
    ZT_HIP(hipMalloc(&buf_dev, (size_t)n * rg * sizeof(u64)));
    hipError_t e = hipMemcpy(ids_dev, ids_host, (size_t)n * sizeof(long long), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(buf_dev, g.data(), (size_t)n * rg * sizeof(u64), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        k_scatter_tppr_rows<<<(unsigned)n, 128>>>(h->rows + (size_t)m * (size_t)h->N * rg, ids_dev, n, (int)rg, buf_dev);
        e = hipDeviceSynchronize();
    }
    (void)hipFree(ids_dev); (void)hipFree(buf_dev);
    ZT_HIP(e);
    h->set[0].valid = h->set[1].valid = false;
    return ZT_OK;
}
