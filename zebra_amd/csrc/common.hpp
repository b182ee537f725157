// Shared host/device helpers for libzebra_amd (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "../../include/zebra_amd.h"

namespace zt {

void set_error(const char *fmt, ...);

// Optional HIP-event timing of individual kernels (zt_profile_* in the ABI):
// an event pair is recorded on the launch stream around the kernel; elapsed
// times are summed per kernel name when the profile is read.
enum ProfId { P_PREPASS = 0, P_STREAM, P_CLEANUP, P_PRUNE, P_EMBED_PREP, P_FC1_AGG, P_EMBED_OUT, P_STORE_MSG, P_GRU,
              P_SCORE, P_EXCHANGE, P_COUNT };
extern bool g_prof_on;
void prof_begin(hipStream_t s, int id);
void prof_end(hipStream_t s, int id);
// zt_tppr_stream for callers inside the library (tppr_stream.hip)
int tppr_stream_ex(zt_tppr *h, const int32_t *nodes_dev, const double *ts_dev, const int64_t *eidx_dev, int64_t B,
                   int32_t n_roles, int32_t emit, int32_t model, int32_t *out_nodes_dev, int32_t *out_eidx_dev,
                   float *out_dt_dev, float *out_w_dev, uint64_t plan_token, void *stream, bool plan_ordered,
                   hipEvent_t *done_out, int32_t sub_B, int32_t *member_done_dev = nullptr);
// member_done_dev (launches over several batches: sub_B > 0): TPPR_MEMBER_WORDS ints, zero when the launch starts.  Word g counts
// the (edge, model) tasks of batch g whose output rows are written -- visible at agent scope when the count is -- so that a
// consumer of batch g's rows need not wait for the end of the launch: word g == B_g * models opens it (pipeline.hip:
// k_member_gate).  A launch that k_count rejected (ZT_ERR_RANGE) writes its empty rows and sets every word to INT_MAX.
constexpr int TPPR_MAX_MEMBERS = 8;         // batches per launch that can be released one by one
constexpr int TPPR_MEMBER_WORDS = TPPR_MAX_MEMBERS + 1;      // (the last word: workgroups of a rejected launch that are through)
// zt_gru_update with the projected-table refresh folded into the GRU kernel (memory_update.hip); wm_p from embed_wm_ptr.
// counter_zeroed: the row counter (first word of the workspace) is zero already; select_done: the row list and the counter
// are filled (pipeline.hip: the message kernel hands its list of winners over), no compaction of flagged ids
// The output layers of an embed call, held back (embed_ex: `defer`) so that gru_update_ex can launch them in ONE kernel with the
// GRU update (k_out_gru, memory_update.hip): the two are independent apart from the memory rows the source path reads --
// the GRU half waits for those reads before it writes (a gate in the GRU workspace whose whole state is device memory: a
// count of source-path units that have their rows and a count of participants that have left; the last one out zeroes both, so
// every launch finds them at zero whatever the host did in between; the wait is bounded and reports to `status` / `latch`).
struct embed_out_deferred {
    bool valid;
    const float *memory;
    long long num_nodes;
    const int *nodes;
    long long N;
    int D, M, hg;
    int form, gx;          // 1: the tiled kernel (k_embed_out); 2: the latency-organised one (k_embed_out2), gx tiles' worth of waves per path and N-tile
    const float *H, *S, *fc2_p, *fc2_b, *fc1s_p, *fc1s_b, *fc2s_p, *fc2s_b;
    float *out;
    int *status;
    int *latch;            // host-mapped word a gate wait that gives up writes ZT_ERR_TIMEOUT to (or NULL); kept across embed_ex calls
};
int embed_out_launch(const embed_out_deferred &d, void *stream);           // aggregate.hip: the held-back layers as a kernel of their own
int gru_update_ex(float *memory_dev, float *last_update_dev, const float *messages_dev, const float *msg_ts_dev,
                  uint8_t *flags_dev, int64_t num_nodes, int32_t D, int32_t msg_dim, const int32_t *ids_dev, int64_t n_ids,
                  const int32_t *n_ids_dev, const zt_gru_weights *wt, void *workspace_dev, int32_t weights_ready,
                  const float *wm_p, float *proj_table, void *stream, bool counter_zeroed = false, bool select_done = false,
                  embed_out_deferred *fuse = nullptr);
// zt_store_messages_range that also zeroes one int (the GRU update's row counter: first word of its workspace)
int store_messages_ex(const float *memory_dev, const float *last_update_dev, const float *efeat_dev, const float *time_w_dev,
                      int64_t num_nodes, int64_t num_edges, int32_t D, int32_t F, int32_t T, const int32_t *src_dev,
                      const int32_t *dst_dev, const double *ts_dev, const int64_t *eidx_dev, int64_t B, int64_t pos_lo,
                      int64_t pos_hi, float *messages_dev, float *msg_ts_dev, uint8_t *flags_dev, int32_t *scratch_dev,
                      int32_t *uniq_ids_dev, int32_t *n_uniq_dev, int32_t *status_dev, int32_t *zero_word_dev, void *stream,
                      bool *zeroed_out = nullptr, bool set_flags = true);
// The wait for ONE batch's T-PPR rows inside a launch that covers several (pipeline.hip, "release by member"; the counter is
// tppr_stream_ex's member_done_dev): word == nullptr: none.  The persistent aggregation kernels take it as arguments and wait
// behind their weight prologue (no packet of its own on the stream); for the others member_gate_launch puts a one-wave kernel
// in front.
struct member_gate {
    const int32_t *word;
    int32_t target;
    int *latch;            // host-mapped word a wait that gives up writes ZT_ERR_TIMEOUT to (or NULL)
};
int member_gate_launch(const member_gate &g, int32_t *status_dev, void *stream);      // aggregate.hip
// zt_embed with an event the stream waits for between the aggregation kernel and the output layer (pipeline.hip: the
// message build; a wait packet right in front of the GRU costs the step ~6 us of command-processor time, here it is
// processed while the aggregation runs)
int embed_ex(const float *memory_dev, const float *efeat_dev, int64_t num_nodes, int64_t num_edges, int32_t D, int32_t F, int32_t T,
             const int32_t *nodes_dev, int64_t N, int32_t M, int32_t k, const int32_t *nbr_dev, const int32_t *eix_dev,
             const float *dt_dev, const float *w_dev, const zt_embed_weights *wt, float *out_dev, void *workspace_dev,
             int32_t *status_dev, const float *proj_table_dev, int32_t weights_ready, void *stream, hipEvent_t mid_wait,
             embed_out_deferred *defer = nullptr, const member_gate *gate = nullptr);
// W_m (the memory columns of fc1, padded to [Dp][Dp]) inside an embed workspace prepared for (N, D, F, T, M, k) (aggregate.hip)
const float *embed_wm_ptr(void *embed_ws, int64_t N, int32_t D, int32_t F, int32_t T, int32_t M, int32_t k);
// the persistent aggregate kernel for wide edge features (aggregate_wide.hip: F = 172, weights resident in LDS)
bool fc1_agg_wide_supported(int D, int F, int T, int k);
size_t fc1_agg_wide_weight_bytes();
void fc1_agg_wide_pack(const float *fc1_w_dev, const float *time_w_dev, const float *fc1_b_dev, float *packed_dev, hipStream_t s);
int fc1_agg_wide_launch(const float *P, const float *efeat, const float *time_w, long long num_nodes, long long num_edges,
                        long long N, int M, int k, const int *nbr, const int *eix, const float *dt, const float *w,
                        const float *packed, const float *b1, float *G, float *S, int *status, int cus, hipStream_t s,
                        const member_gate *gate = nullptr);
int pruned_topk_multi_fill(const zt_csr *c, const int32_t *q_nodes_dev, const double *q_ts_dev, int64_t nq, int32_t width,
                           int32_t depth, int32_t n_models, const double *alpha_host, const double *beta_host, int32_t k,
                           int32_t *out_nodes_dev, int32_t *out_eidx_dev, float *out_dt_dev, float *out_w_dev,
                           int32_t *status_dev, void *stream);     // tppr_prune.hip: empty rows written as zeros (no memset before)
void tppr_hint_cus(zt_tppr *h, hipStream_t s);      // tppr_prepass.hip: plan for the compute units of stream s
// exchange.hip: the row exchange of one step of a multi-GPU run on `stream` (pack -> all-gather -> scatter); the ids of
// the rows written (-1: padding) come back for the projected-row refresh
int exchange_step(zt_exchange *x, const int32_t *rows_dev, const int32_t *count_dev, void *stream, const int32_t **ids_out,
                  int64_t *n_ids_out);
void exchange_shape(const zt_exchange *x, int *rank, int *world);
constexpr int TPPR_MAX_LAUNCH = 16384;     // edges one T-PPR launch can cover (tppr_stream.hip: MAX_CHUNK)

// flags of the events that only order streams of this device
inline unsigned sync_event_flags() { return 0u; }
// zt_set_kernel_choice: 0 = the library picks by shape (runtime.hip)
int kernel_choice(int which);
// alternatives that were measured slower are compiled into variant builds only (tools/build_variant.sh): what THIS build has
bool tppr_chain_mode_compiled(int mode);       // tppr_stream.hip
bool tppr_prepass_coop_compiled();             // tppr_prepass.hip
// CUs a stream may use (CU-masked streams: the size of the mask; queried per call -- runtime.hip)
int stream_cu_count(hipStream_t s);
#define ZT_PROF_BEGIN(s, id) do { if (zt::g_prof_on) zt::prof_begin((s), (id)); } while (0)
#define ZT_PROF_END(s, id) do { if (zt::g_prof_on) zt::prof_end((s), (id)); } while (0)

#define ZT_HIP(expr)                                                                          \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess) {                                                              \
            zt::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__,   \
                          __LINE__);                                                          \
            return ZT_ERR_HIP;                                                                \
        }                                                                                     \
    } while (0)

#define ZT_LAUNCH_CHECK()                                                                     \
    do {                                                                                      \
        hipError_t e__ = hipGetLastError();                                                   \
        if (e__ != hipSuccess) {                                                              \
            zt::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__),         \
                          __FILE__, __LINE__);                                                \
            return ZT_ERR_HIP;                                                                \
        }                                                                                     \
    } while (0)

typedef unsigned long long u64;

constexpr int WAVE = 64;

// ---- device helpers ---------------------------------------------------------
#if defined(__HIPCC__)

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }

// Make this wave's LDS writes visible to its own later cross-lane LDS reads
// (orders the compiler and drains lgkmcnt; no workgroup barrier involved).
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// v from the lane a DPP control selects (0x120 | n: row_ror:n -- rotation inside each row of 16 lanes)
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}

__device__ __forceinline__ u64 lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// Agent-scope relaxed accesses: lower to global_load/store ... sc1 (L1 bypass /
// write-through), the form cross-CU hand-offs inside one launch need
// (cdna_hip_programming.md Guideline 16, recipe R1).
__device__ __forceinline__ u64 ld_agent(const u64 *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(u64 *p, u64 v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_agent(const double *p)
{
    return __longlong_as_double((long long)ld_agent(reinterpret_cast<const u64 *>(p)));
}
__device__ __forceinline__ void st_agent(double *p, double v)
{
    st_agent(reinterpret_cast<u64 *>(p), (u64)__double_as_longlong(v));
}
__device__ __forceinline__ int ld_agent(const int *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(int *p, int v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(float *p, float v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned ld_agent(const unsigned *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(unsigned *p, unsigned v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ONE lane's atomic add on behalf of its wave, without a branch in the program: the wave's queue claims (k_stream).
// What the obvious forms compile to (seen in the ISA, round 6): `if (lane == 0) old = atomicAdd(p, 1)` is jump-threaded with
// lane-0 code around it and the structurizer replays the loop body for the other lanes; `atomicAdd(p, lane == 0 ? 1 : 0)` --
// written to avoid that -- makes the compiler's atomic optimizer wrap the add in a scan over the 64 lanes ONE LANE AT A TIME
// (s_ff1 / v_readlane / v_writelane / s_andn2 ...: 9 instructions x 64 = ~580 per claim, on a chain workgroup whose instruction
// issue is what bounds a hub chain); `atomicAdd(p, 1)` by every lane with a pointer the compiler cannot prove uniform is 64
// atomics to one address (~10 ns each at the memory side).  Here: exec = lane 0 for the one instruction.
// PRECONDITION: called in wave-uniform control flow with every lane active (the top of k_stream's two queue loops) -- lane 0
// issues the add with ITS copy of the operands, and the result is read from lane 0.
// Returns the value before the add, wave-uniform.  p: LDS.
__device__ __forceinline__ int wave_claim_lds(int *p, int v)
{
    int old;
    unsigned long long save;
    const unsigned a = (unsigned)(unsigned long long)p;              // generic -> LDS: the low 32 bits are the LDS address
    asm volatile("s_mov_b64 %1, exec\n\t"
                 "s_mov_b64 exec, 1\n\t"
                 "ds_add_rtn_u32 %0, %2, %3\n\t"
                 "s_mov_b64 exec, %1\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(old), "=&s"(save) : "v"(a), "v"(v) : "memory");
    return __builtin_amdgcn_readlane(old, 0);
}
// p: global memory (agent scope, as atomicAdd)
__device__ __forceinline__ int wave_claim_global(int *p, int v)
{
    int old;
    unsigned long long save;
    asm volatile("s_mov_b64 %1, exec\n\t"
                 "s_mov_b64 exec, 1\n\t"
                 "global_atomic_add %0, %2, %3, off sc0\n\t"
                 "s_mov_b64 exec, %1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(old), "=&s"(save) : "v"(p), "v"(v) : "memory");
    return __builtin_amdgcn_readlane(old, 0);
}
// ... and without a result: fire and forget
__device__ __forceinline__ void wave_add_global(int *p, int v)
{
    unsigned long long save;
    asm volatile("s_mov_b64 %0, exec\n\t"
                 "s_mov_b64 exec, 1\n\t"
                 "global_atomic_add %1, %2, off\n\t"
                 "s_mov_b64 exec, %0"
                 : "=&s"(save) : "v"(p), "v"(v) : "memory");
}

// by ONE thread: poll `word` (agent scope) until it reaches `target`; bounded like every in-kernel wait of the library (4 s of
// the 100 MHz wall clock), then ZT_ERR_TIMEOUT to the status word and the latch.  false: gave up.
// SLEEP: 64-clock units between two polls.  Inside the persistent aggregation kernels EVERY workgroup polls the one word:
// at ~0.3 us a poll, 192 of them kept one memory channel busy with nothing else and k_stream -- whose counters live in that
// line -- lost 5 % (measured); there a poll every ~4 us (a gate that is open when the kernel arrives costs no sleep at all).
template <int SLEEP = 8>
__device__ __forceinline__ bool member_gate_wait(const int *word, int target, int *status, int *latch)
{
    unsigned spins = 0;
    long long t0 = 0;
    while (ld_agent(word) < target) {
        __builtin_amdgcn_s_sleep(SLEEP);
        if ((++spins & (SLEEP >= 64 ? 127u : 1023u)) == 0) {
            const long long now = (long long)wall_clock64();
            if (t0 == 0) t0 = now;
            else if (now - t0 > 400000000ll) {
                atomicExch(status, ZT_ERR_TIMEOUT);
                if (latch != nullptr) __hip_atomic_store(latch, (int)ZT_ERR_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                return false;
            }
        }
    }
    return true;
}
// a whole workgroup: thread 0 waits, the others wait for it (word == nullptr: nothing, not even the barrier)
__device__ __forceinline__ void member_gate_enter(const int *word, int target, int *status, int *latch)
{
    if (word == nullptr) return;
    if (threadIdx.x == 0) (void)member_gate_wait<127>(word, target, status, latch);
    __syncthreads();
}

// cos(x) for the time encoding.  |x| < 4e6: float32 Cody-Waite reduction with three FMA steps
// (k = rint(x*2/pi), pi/2 split into floats, two FMA steps: each is exact up to one rounding of an O(1)
// value); larger arguments: the same reduction in float64.  Float32 polynomials on [-pi/4, pi/4].
// |err| <= 3e-7 over |x| <= 3e8 (ocml cosf: 7e-8), at a fraction of the instructions of the
// Payne-Hanek path large arguments take in ocml.
__device__ __forceinline__ float time_cos_poly(float rf, int q)
{
    const float s = rf * rf;
    const float c = fmaf(s, fmaf(s, fmaf(s, fmaf(s, 2.4801587e-5f, -1.3888889e-3f), 4.1666668e-2f), -0.5f), 1.0f);
    const float sn = rf * fmaf(s, fmaf(s, fmaf(s, fmaf(s, 2.7557319e-6f, -1.9841270e-4f), 8.3333338e-3f), -1.6666667e-1f), 1.0f);
    const float res = (q & 1) ? sn : c;
    return ((q + 1) & 2) ? -res : res;
}

__device__ __noinline__ float time_cosf_large(float x)
{
    // beyond 1e15 the quotient k no longer fits the float64 reduction below (k * pi/2 must stay exact to
    // ~1e-17 * k): take the library's Payne-Hanek path; such arguments (dt > 3e7 years) are not data
    if (!(fabsf(x) < 1.0e15f)) return cosf(x);
    const double xd = (double)x;
    const double kd = rint(xd * 0.63661977236758134308);
    double r = fma(-kd, 1.57079632679489655800e+00, xd);
    r = fma(-kd, 6.12323399573676603587e-17, r);
    return time_cos_poly((float)r, (int)((long long)kd & 3ll));   // |k| < 6.4e14: the quadrant from 64 bits
}

// the |x| < 4e6 formula of time_cosf alone (callers check the range themselves)
__device__ __forceinline__ float time_cosf_fast(float x)
{
    const float kf = rintf(x * 0.636619772f);
    float r = fmaf(kf, -1.57079637e+00f, x);
    r = fmaf(kf, 4.37113883e-08f, r);
    // quarter revolutions of the quadrant: fract(k / 4) = (k & 3) / 4, also for negative k
    return __builtin_amdgcn_cosf(fmaf(r, 0.159154943f, __builtin_amdgcn_fractf(kf * 0.25f)));
}

__device__ __forceinline__ float time_cosf(float x)
{
    if (fabsf(x) >= 4.0e6f) return time_cosf_large(x);
    const float kf = rintf(x * 0.636619772f);
    float r = fmaf(kf, -1.57079637e+00f, x);            // pi/2 = 1.57079637 - 4.37113883e-08 - 1.7e-15 (floats);
    r = fmaf(kf, 4.37113883e-08f, r);                   // the third term is < 5e-9 for |k| < 2.6e6
    // |r| <= pi/4 exactly reduced; the quadrant goes back in as quarter revolutions and v_cos_f32
    // (argument in revolutions) finishes: 9 instructions, max |err| 3e-7 over |x| < 4e6 (measured
    // against float64 cos on 4M arguments: a round-2 microbenchmark)
    const int q = (int)kf & 3;
    return __builtin_amdgcn_cosf(fmaf(r, 0.159154943f, 0.25f * (float)q));
}

// (aggregate.hip, aggregate_wide.hip: the time encoding computed in the MFMA operand lanes)
__device__ __forceinline__ float time_cosf_rev(float x)
{
    // cos(x) for |x| up to 3e8 (dt is seconds: 10 years) in SIX float32 instructions -- on this chip an f32 MFMA and
    // vector instructions of the same SIMD do not overlap, so every vector instruction beside the MFMAs costs its
    // full issue time (float64 ones twice that).  Revolutions x / 2pi with 1 / 2pi = c1 + c2 (two floats): the
    // product x c1 and its rounding error (one FMA) are carried separately, so the FRACTION of the large term is
    // exact; x c2 < 2 needs no such care.  rev = fract(x c1) + (err1 + x c2), then v_cos_f32 (argument in
    // revolutions).  max |error| 9e-7 over |x| <= 3e8 (mean 7e-8): tools/exp/cos_rev_check.py.
    const float c1 = 0x1.45f306p-3f, c2 = 0x1.b9391p-28f;
    const float p1 = x * c1, e1 = fmaf(x, c1, -p1);
    return __builtin_amdgcn_cosf(__builtin_amdgcn_fractf(p1) + fmaf(x, c2, e1));
}

// Training dropout of the hidden layer (nn.Dropout(0.1) between fc1's ReLU and fc2, modules/embedding_module.py:89,
// 323-326): the keep-mask of element idx = (gathered row) * D + column is a hash of (seed, idx), so the backward
// kernel regenerates it instead of reading it back.  thr = p * 2^32; returns 1 / (1 - p) for a kept element, else 0.
// (zebra_amd/modules.py: dropout_mask is the same arithmetic in numpy, for the tests.)
__device__ __forceinline__ float drop_scale(unsigned seed_lo, unsigned seed_hi, unsigned thr, float inv_keep,
                                            unsigned long long idx)
{
    unsigned h = ((unsigned)idx * 0x9E3779B1u) ^ seed_lo;
    h ^= ((unsigned)(idx >> 32) * 0x85EBCA77u) + seed_hi;
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;      // murmur3 finaliser
    return h >= thr ? inv_keep : 0.f;
}

#endif  // __HIPCC__
// p -> threshold of drop_scale (0 = no dropout); host side
inline unsigned drop_threshold(float p)
{
    if (!(p > 0.f)) return 0u;
    const double t = (double)p * 4294967296.0;
    return t >= 4294967295.0 ? 4294967295u : (unsigned)t;
}
#if defined(__HIPCC__)
// floor(f / d) for small f via a precomputed multiplier m = fastdiv_magic(d); exact for f*d < 2^32.
__device__ __forceinline__ unsigned fastdiv_magic(unsigned d) { return d <= 1u ? 0u : 0xFFFFFFFFu / d + 1u; }   // 0 = divide by 1
__device__ __forceinline__ int fastdiv(int f, unsigned m) { return m == 0u ? f : (int)__umulhi((unsigned)f, m); }

#endif  // __HIPCC__

}  // namespace zt
