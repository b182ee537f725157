// Library runtime: error text, per-kernel HIP-event timing (zt_profile_*), version, CU-masked streams.
#include "common.hpp"

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
                                           "embed_prep", "fc1_agg", "embed_out", "store_messages", "gru_update", "score",
                                           "exchange"};
hipEvent_t prof_event()
{
    hipEvent_t e;
    if (!g_prof_pool.empty()) { e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
int g_prof_every = 1;                 // time every n-th launch of each kernel (zt_profile_enable(n))
long long g_prof_seen[P_COUNT];
long long g_prof_open[P_COUNT] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};   // index of the open record per kernel
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


// ---- kernel selection overrides (zt_set_kernel_choice) ---------------------------------
namespace { int g_choice[ZT_CHOICE_COUNT] = {0}; }
int kernel_choice(int which) { return (which >= 0 && which < ZT_CHOICE_COUNT) ? g_choice[which] : 0; }

// CUs the stream may use (CU-masked streams: the size of the mask).  Queried per call: a cache keyed by the stream
// handle goes stale when a pipeline is destroyed and the runtime hands the same handle value to a stream with another
// mask (round-3 advisor); the query is microseconds next to the kernels it sizes.
int stream_cu_count(hipStream_t s)
{
    static int total = 0;
    int tot = total;
    if (tot == 0) {
        hipDeviceProp_t prop;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        tot = prop.multiProcessorCount;
        total = tot;
    }
    uint32_t mask[32] = {0};
    int c = tot;
    if (hipExtStreamGetCUMask(s, 32, mask) == hipSuccess) {
        int n = 0;
        for (int q = 0; q < 32; ++q) n += __builtin_popcount(mask[q]);
        if (n > 0 && n < tot) c = n;
    } else {
        (void)hipGetLastError();
    }
    return c;
}

}  // namespace zt

using namespace zt;

extern "C" int zt_set_kernel_choice(int32_t which, int32_t value)
{
    if (which < 0 || which >= ZT_CHOICE_COUNT || value < 0) { set_error("zt_set_kernel_choice: unknown selector %d", which); return ZT_ERR_ARG; }
    // alternatives that were measured slower than the library's pick live in variant builds only (tools/build_variant.sh,
    // tools/exp/variants/): this build says so instead of silently running its default
    if ((which == ZT_CHOICE_TPPR_CHAIN && !zt::tppr_chain_mode_compiled(value)) ||
        (which == ZT_CHOICE_TPPR_PREPASS && value == ZT_PREPASS_COOP && !zt::tppr_prepass_coop_compiled())) {
        set_error("zt_set_kernel_choice: value %d of selector %d is compiled into variant builds only (tools/build_variant.sh)", value, which);
        return ZT_ERR_UNSUPPORTED;
    }
    g_choice[which] = value;
    return ZT_OK;
}

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

