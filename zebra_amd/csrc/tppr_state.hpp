// The streaming T-PPR handle: row layout constants, struct zt_tppr (device rows, two sets of prepass buffers, launch
// epoch, failure latch) and the host helpers shared by the translation units of the path:
//   tppr_prepass.hip  the dependency prepass (kernels + plan_chunk + zt_tppr_plan)
//   tppr_stream.hip   k_stream (tppr_rows.hpp: rows and merges; tppr_hop.hpp: the general hop; tppr_chain.hpp: the hub
//                     chain's hop) + create / destroy / reset / copy / run
//   tppr_io.hip       export / import of dictionaries
#pragma once

#include "common.hpp"

#include <cstring>

using zt::u64;

// ---- handle --------------------------------------------------------------------
namespace {
constexpr int HDR = 4;                 // header granules: len, norm lo, norm hi, pad
constexpr int ORD_BITS = 15;           // writer ordinal inside one launch (a node has at most MAX_CHUNK writers)
constexpr int MAX_CHUNK = 16384;       // edges per launch (ordinals must fit ORD_BITS)
constexpr unsigned EPOCH_MAX = (1u << (32 - ORD_BITS)) - 1;
static_assert(MAX_CHUNK == zt::TPPR_MAX_LAUNCH, "common.hpp: TPPR_MAX_LAUNCH");
// hub chains: the nodes touched most often in a launch get a workgroup of their own
// (overridable for experiments: tools/exp/chains_exp.sh -- 32 / 48 chains and a threshold of 12 were measured in round 4:
//  no gain on C3, a loss on C5, where chain workgroups then crowd out the general queue)
#ifndef ZT_HOT_MIN
#define ZT_HOT_MIN 24
#endif
#ifndef ZT_MAX_CHAINS
#define ZT_MAX_CHAINS 16
#endif
constexpr int HOT_MIN = ZT_HOT_MIN;       // accesses in one launch that make a node a chain candidate
constexpr int MAX_HOT = 128;           // candidates kept
constexpr int BIG_MIN = 48;            // a node group of this many accesses gets its dependencies from a cooperative sort
constexpr int MAX_BIG = 512;           // (tppr_prepass.hip: d_deps_group); the list of such nodes sits behind hot_node[MAX_HOT]
constexpr int DEPS_SORT_MAX = 4096;    // members of a group the sort takes (larger groups: the per-access loop)
constexpr int MAX_CHAINS = ZT_MAX_CHAINS; // chains per model
#ifdef ZT_WAITLOG
constexpr int CTL_LOG = 60;            // (diagnostic build: tools/build_waitlog.sh)
#else
constexpr int CTL_LOG = 6;             // timeout reports kept per launch
#endif
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
                     // (dictionary order, tagged with the launch epoch), or nullptr (k > REG_K_MAX: no chains).  Within a
                     // launch a hub's row lives HERE: everybody but its own chain reads it by version, and only the
                     // chain's last hop stores it back to `rows`
    double *hubscale;  // [M][MAX_CHAINS][CH_MAX + 1][4]: {norm, norm * beta + beta, scale_s1, scale_s2} of a chain's hub at every
                       // position (utils/util.py:519-522).  A node's norm moves by norm <- norm * beta + beta with every edge that
                       // touches it, whatever the rows hold: the whole sequence follows from the norm the launch finds, and the
                       // chain workgroup works it out ONCE, all positions in parallel, when it starts (k_stream) -- a hop loads four
                       // doubles instead of iterating the recurrence and dividing twice (~100 instructions of a compute unit whose
                       // instruction issue is what bounds a chain: round 6).  Allocated with hubver.
    // hub chains of the launch
    int *chain_of;     // [N] chain index of a hub node, -1 otherwise (all -1 between calls)
    int *hot_node;     // [MAX_HOT] candidates, hot_cnt their access counts; [MAX_HOT .. MAX_HOT + MAX_BIG): nodes of big groups
    int *hot_cnt;
    int *chain_node;   // [MAX_CHAINS]
    int *chain_len;    // [MAX_CHAINS]
    int *chain_edges;  // [MAX_CHAINS][CH_MAX] the hub's edges in order: position = the hub's writer ordinal (wo) at that edge.
                       // An edge between two hubs is in BOTH chains: each applies its own hub's update
    int *owner_of;     // [MAX_CHUNK] the chain whose partner task (general queue) emits the edge's rows, or -1
    int *hv;           // [3 * MAX_CHUNK] per access: the chain that holds the accessed node's row by version (the row to
                       // read is version wo of that chain), or -1: the row is read from / written to `rows`
    // control words (device): [0] cursor, [1] queue head, [2] status, [3] hot candidates, [4] chains, [5] big groups,
    // [13] timeout reports, [16..] the reports (see note_timeout)
    int *ctl;
    unsigned epoch;
    int n_cu;
    int run_cus;     // CUs of the stream the last k_stream ran on (0: not known yet; zt_pipeline_create tells ahead of the first plan)
    int share;       // processes whose T-PPR kernels share this device's CUs (zt_tppr_set_device_share; 1 = the device is ours)
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
            *chain_edges, *owner_of, *hv, *ctl;
        hipEvent_t planned, consumed;      // prepass finished / k_stream finished with the set
        bool used;                         // `consumed` has been recorded at least once
        // what the set was planned for (valid == a zt_tppr_plan result not consumed yet)
        bool valid;
        const int32_t *nodes;
        int B, n_roles, model, grid, max_chains;
        unsigned long long token;
        unsigned bar_base;                 // k_prepass_coop: value of the set's barrier word (ctl[11], never reset) before the next launch
    } set[2];
    int next_set;
    // last launch (diagnostics)
    const int *dbg_nodes;
    long long dbg_stride;
    int dbg_B, dbg_roles, dbg_models;
};

namespace {
__host__ __device__ inline u64 granule(unsigned tag, unsigned payload) { return ((u64)tag << 32) | payload; }

// point the handle's prepass fields at one of its two sets
inline void use_set(zt_tppr *h, int q)
{
    const zt_tppr::PlanSet &P = h->set[q];
    h->cnt = P.cnt; h->off = P.off; h->slot = P.slot; h->list = P.list; h->wo = P.wo; h->pflag = P.pflag; h->nxt = P.nxt;
    h->chain_of = P.chain_of; h->hot_node = P.hot_node; h->hot_cnt = P.hot_cnt; h->chain_node = P.chain_node;
    h->chain_len = P.chain_len; h->chain_edges = P.chain_edges; h->owner_of = P.owner_of; h->hv = P.hv; h->ctl = P.ctl;
}

// a failure latched by an earlier launch makes every later call fail until zt_tppr_status has reported it
inline int latched(const zt_tppr *h, const char *who)
{
    const int st = *reinterpret_cast<volatile int *>(h->latch_host);
    if (st == 0) return ZT_OK;
    zt::set_error("%s: an earlier launch on this handle failed (%s); zt_tppr_status reports and clears it", who,
                  st == ZT_ERR_RANGE ? "node or edge id out of range, that batch was not applied" : "dependency wait timed out");
    return st;
}

#if defined(__HIPCC__)
// first failure of a launch -> the handle's host-mapped latch (system scope: the host reads it without a synchronisation)
__device__ __forceinline__ void latch_failure(int *latch, int code)
{
    __hip_atomic_store(latch, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
#endif
}  // namespace

namespace zt {
constexpr int TPPR_WAVES_PER_WG = 8;       // waves of a k_stream workgroup (a hub chain's waves)
constexpr int TPPR_REG_K_MAX = 30;         // the register-resident merge and the hub chains serve k <= 30 (tppr_rows.hpp)
// tppr_prepass.hip
int tppr_stream_cus(const zt_tppr *h, hipStream_t s);
int tppr_chains_for_grid(const zt_tppr *h, int grid, int n_models);
void tppr_hint_cus(zt_tppr *h, hipStream_t s);
void tppr_launch_shape(const zt_tppr *h, int cus, int B, int n_models, int *grid_out, int *max_chains_out);
int tppr_plan_chunk(zt_tppr *h, int q, const int32_t *nodes, const long long *eidx, long long role_stride, int B, int n_roles,
                    int model, hipStream_t s);
}  // namespace zt
