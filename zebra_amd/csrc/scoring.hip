// Link scoring and link-prediction metrics on the device.
//
// zt_affinity replaces TGN.compute_edge_probabilities' scorer (reference model/tgn_model.py:185-188 with
// MergeLayer, utils/util.py:14-26):
//   score = fc2(relu(fc1(cat[x1, x2]))),  x1 = [src | src], x2 = [dst | neg]  ->  sigmoid
// for the B positive pairs (src, dst) and the B negative pairs (src, neg) of a batch whose embeddings
// [3B][H] = [src | dst | neg] the aggregation has just written.  fc1 acts on a concatenation, so
// fc1([a | b]) = W_a a + W_b b + b1 and W_a src is SHARED by an edge's two pairs: three [16 x H] x [H x 16] products
// per 16 edges and N-tile instead of the reference's four.
//
// Organisation (the batch is small, the chain of dependent memory round trips is what costs): one WAVE per N-tile of
// the hidden layer keeps its 16 columns of W_a and W_b in registers and strides over tiles of 16 edges; the A operand
// comes straight from the embedding rows into the MFMA lanes (float4 at columns 16 c + 4 g: the k order the padded
// weights have); bias, ReLU and fc2's weight are applied in the accumulator lanes and summed over the wave's 16 columns
// with four DPP rotations.  The N-tiles' partial scores meet in memory: the LAST wave to arrive for a tile (one counter
// per tile, reset by that wave) adds them first to last -- a fixed order, so the probabilities do not depend on which
// wave finishes when -- and applies fc2's bias and the sigmoid.
//
// zt_link_metrics replaces the per-batch scikit-learn calls of evaluation/evaluation.py:34-45 and train.py:218-227
// (average_precision_score, roc_auc_score, accuracy of argmax) with ONE single-workgroup kernel: bitonic sort of the 2B
// scores in LDS, two scans (true positives; start of every run of equal scores) and the two curve sums over the distinct
// thresholds in float64, ties handled as scikit-learn does (zebra_amd/evaluation.py states the definitions).
#include "common.hpp"

using namespace zt;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline int round_up16(int x) { return (x + 15) / 16 * 16; }

// fc1.weight [H][2H], fc1.bias [H], fc2.weight [1][H]  ->  Wa_p, Wb_p [Hp][Hp] (zero padded), b1_p [Hp], w2_p [Hp]
__global__ void k_pack_affinity(const float *__restrict__ fc1_w, const float *__restrict__ fc1_b, const float *__restrict__ fc2_w,
                                int H, int Hp, float *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int mat = Hp * Hp;
    if (i < 2 * mat) {
        const int which = i / mat, r = (i % mat) / Hp, c = i % Hp;
        out[i] = (r < H && c < H) ? fc1_w[(size_t)r * 2 * H + which * H + c] : 0.f;
    } else if (i < 2 * mat + Hp) {
        const int c = i - 2 * mat;
        out[i] = c < H ? fc1_b[c] : 0.f;
    } else if (i < 2 * mat + 2 * Hp) {
        const int c = i - 2 * mat - Hp;
        out[i] = c < H ? fc2_w[c] : 0.f;
    }
}

template <int KC>
__global__ __launch_bounds__(64) void k_affinity(const float *__restrict__ emb, long long B, int H, const float *__restrict__ packed,
                                                 const float *__restrict__ fc2_b, float *part, int *counters, float *__restrict__ prob)
{
    constexpr int Hp = KC * 16, NT = KC;
    const int lane = threadIdx.x, r16 = lane & 15, g4 = lane >> 4;
    const int nt = blockIdx.y, col = 16 * nt + r16;
    const float *Wa = packed, *Wb = packed + (size_t)Hp * Hp, *b1p = Wb + (size_t)Hp * Hp, *w2p = b1p + Hp;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 wa[KC], wb[KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        wa[c] = *reinterpret_cast<const f32x4 *>(Wa + (size_t)col * Hp + 16 * c + 4 * g4);
        wb[c] = *reinterpret_cast<const f32x4 *>(Wb + (size_t)col * Hp + 16 * c + 4 * g4);
    }
    const float b1v = b1p[col], w2v = w2p[col], b2v = fc2_b[0];
    bool cin[KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) cin[c] = 16 * c + 4 * g4 < H;           // H % 4 == 0: all four columns or none
    const long long tiles = (B + 15) / 16, Bp = tiles * 16;
    for (long long t = blockIdx.x; t < tiles; t += gridDim.x) {
        const long long e0 = t * 16;
        const bool rin = e0 + r16 < B;
        const float *ps = emb + (size_t)(rin ? e0 + r16 : 0) * H + 4 * g4, *pd = ps + (size_t)B * H, *pn = pd + (size_t)B * H;
        f32x4 as[KC], ad[KC], an[KC];
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const bool ok = rin && cin[c];
            as[c] = ok ? *reinterpret_cast<const f32x4 *>(ps + 16 * c) : zero4;
            ad[c] = ok ? *reinterpret_cast<const f32x4 *>(pd + 16 * c) : zero4;
            an[c] = ok ? *reinterpret_cast<const f32x4 *>(pn + 16 * c) : zero4;
        }
        f32x4 au = zero4, ap = zero4, ang = zero4;
#pragma unroll
        for (int c = 0; c < KC; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                au = __builtin_amdgcn_mfma_f32_16x16x4f32(as[c][j], wa[c][j], au, 0, 0, 0);
                ap = __builtin_amdgcn_mfma_f32_16x16x4f32(ad[c][j], wb[c][j], ap, 0, 0, 0);
                ang = __builtin_amdgcn_mfma_f32_16x16x4f32(an[c][j], wb[c][j], ang, 0, 0, 0);
            }
        // lane (column col, edges 4 g4 + j): relu(fc1) x fc2's weight, summed over the wave's 16 columns
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float hp = au[j] + ap[j] + b1v, hn = au[j] + ang[j] + b1v;
            hp = (hp > 0.f ? hp : 0.f) * w2v;
            hn = (hn > 0.f ? hn : 0.f) * w2v;
            hp += dpp_f<0x128>(hp); hp += dpp_f<0x124>(hp); hp += dpp_f<0x122>(hp); hp += dpp_f<0x121>(hp);
            hn += dpp_f<0x128>(hn); hn += dpp_f<0x124>(hn); hn += dpp_f<0x122>(hn); hn += dpp_f<0x121>(hn);
            if (r16 == 0) {                                              // (lane 0 of the row: ITS association of the 16 terms)
                part[((size_t)nt * 2 + 0) * Bp + e0 + 4 * g4 + j] = hp;
                part[((size_t)nt * 2 + 1) * Bp + e0 + 4 * g4 + j] = hn;
            }
        }
        __threadfence();
        int done = 0;
        if (lane == 0) done = atomicAdd(&counters[t], 1);
        done = __builtin_amdgcn_readfirstlane(done);
        if (done == NT - 1) {                                            // the last N-tile of this tile to arrive
            __threadfence();
            if (lane < 32) {
                const int e = lane & 15, which = lane >> 4;
                float pv[NT];                                            // all of them requested before the first is looked at
#pragma unroll
                for (int q = 0; q < NT; ++q)
                    pv[q] = __int_as_float(ld_agent(reinterpret_cast<const int *>(part + ((size_t)q * 2 + which) * Bp + e0 + e)));
                float sc = 0.f;
#pragma unroll
                for (int q = 0; q < NT; ++q) sc += pv[q];
                sc += b2v;
                if (e0 + e < B) prob[(size_t)which * B + e0 + e] = 1.f / (1.f + expf(-sc));
            }
            if (lane == 0) atomicExch(&counters[t], 0);                 // ready for the next launch
        }
    }
}

// Large batches (B > 2048: C5).  The wave-per-N-tile kernel above reads every embedding row once per N-tile (19 x) and
// runs one wave per SIMD: 190 us at B = 4096.  Here a workgroup of four waves owns 32 edges: their 96 embedding rows are
// staged in LDS once (118 KB, every load in flight before the first store), each wave takes every fourth N-tile with all
// six row tiles in accumulators (120 registers) and streams its own slice of W_a / W_b from L2 a k-chunk ahead -- each
// weight is read once per workgroup, each embedding row once per launch.  The waves' partial scores meet in LDS and
// are added in wave order.
template <int KC, int ET>           // ET: edges per workgroup, 16 (mid-size batches: more workgroups) or 32
__global__ __launch_bounds__(256, ET == 16 ? 2 : 1) void k_affinity_tiled(const float *__restrict__ emb, long long B, int H, const float *__restrict__ packed,
                                                           const float *__restrict__ fc2_b, float *__restrict__ prob)
{
    constexpr int Hp = KC * 16, NT = KC, NQ = (NT + 3) / 4, ldA = Hp + 4, ROWS = 3 * ET, H4MAX = Hp / 4, EM = ET / 16, MTS = 3 * EM;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *A = reinterpret_cast<float *>(smem);                          // [3 ET][ldA]: src rows, dst rows, neg rows of the tile
    float *part = A + ROWS * ldA;                                        // [4 waves][2][ET]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g4 = lane >> 4;
    const float *Wa = packed, *Wb = packed + (size_t)Hp * Hp, *b1p = Wb + (size_t)Hp * Hp, *w2p = b1p + Hp;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const long long e0 = (long long)blockIdx.x * ET;
    const int h4 = H / 4;                                                // float4 per row (H % 4 == 0)
    // ---- stage the 96 rows: zero beyond the batch and beyond H ----
    constexpr int NST = (ROWS * H4MAX + 255) / 256;
    f32x4 st[NST];
#pragma unroll
    for (int q = 0; q < NST; ++q) {
        const int f = tid + q * 256, row = f / H4MAX, c4 = f - row * H4MAX;
        const long long e = e0 + (row & (ET - 1));
        const bool ok = row < ROWS && c4 < h4 && e < B;
        st[q] = ok ? *reinterpret_cast<const f32x4 *>(emb + ((size_t)(row / ET) * B + e) * H + 4 * c4) : zero4;
    }
    // this wave's first weight chunk rides with the staging loads
    bool live[NQ];
    int colq[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) { live[q] = wave + 4 * q < NT; colq[q] = 16 * (live[q] ? wave + 4 * q : 0) + r16; }
    f32x4 wa[2][NQ], wb[2][NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        wa[0][q] = *reinterpret_cast<const f32x4 *>(Wa + (size_t)colq[q] * Hp + 4 * g4);
        wb[0][q] = *reinterpret_cast<const f32x4 *>(Wb + (size_t)colq[q] * Hp + 4 * g4);
    }
#pragma unroll
    for (int q = 0; q < NST; ++q) {
        const int f = tid + q * 256, row = f / H4MAX, c4 = f - row * H4MAX;
        if (row < ROWS) *reinterpret_cast<f32x4 *>(A + row * ldA + 4 * c4) = st[q];
    }
    __syncthreads();
    f32x4 acc[MTS][NQ];
#pragma unroll
    for (int mt = 0; mt < MTS; ++mt)
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[mt][q] = zero4;
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        if (c + 1 < KC) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                wa[(c + 1) & 1][q] = *reinterpret_cast<const f32x4 *>(Wa + (size_t)colq[q] * Hp + 16 * (c + 1) + 4 * g4);
                wb[(c + 1) & 1][q] = *reinterpret_cast<const f32x4 *>(Wb + (size_t)colq[q] * Hp + 16 * (c + 1) + 4 * g4);
            }
        }
        f32x4 a[MTS];
#pragma unroll
        for (int mt = 0; mt < MTS; ++mt) a[mt] = *reinterpret_cast<const f32x4 *>(A + (16 * mt + r16) * ldA + 16 * c + 4 * g4);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (!live[q]) continue;
#pragma unroll
                for (int mt = 0; mt < MTS; ++mt)
                    acc[mt][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][j], mt < EM ? wa[c & 1][q][j] : wb[c & 1][q][j], acc[mt][q], 0, 0, 0);
            }
    }
    // ---- relu(fc1) x fc2's weight over this wave's columns: lane (column, edges 16 et + 4 g4 + j) ----
    float sp[EM][4], sn[EM][4];
#pragma unroll
    for (int et = 0; et < EM; ++et)
#pragma unroll
        for (int j = 0; j < 4; ++j) { sp[et][j] = 0.f; sn[et][j] = 0.f; }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if (!live[q]) continue;
        const float b1v = b1p[colq[q]], w2v = w2p[colq[q]];
#pragma unroll
        for (int et = 0; et < EM; ++et)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float hp = acc[et][q][j] + acc[EM + et][q][j] + b1v, hn = acc[et][q][j] + acc[2 * EM + et][q][j] + b1v;
                sp[et][j] += (hp > 0.f ? hp : 0.f) * w2v;
                sn[et][j] += (hn > 0.f ? hn : 0.f) * w2v;
            }
    }
#pragma unroll
    for (int et = 0; et < EM; ++et)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float x = sp[et][j], y = sn[et][j];
            x += dpp_f<0x128>(x); x += dpp_f<0x124>(x); x += dpp_f<0x122>(x); x += dpp_f<0x121>(x);
            y += dpp_f<0x128>(y); y += dpp_f<0x124>(y); y += dpp_f<0x122>(y); y += dpp_f<0x121>(y);
            if (r16 == 0) { part[(wave * 2 + 0) * ET + 16 * et + 4 * g4 + j] = x; part[(wave * 2 + 1) * ET + 16 * et + 4 * g4 + j] = y; }
        }
    __syncthreads();
    if (tid < 2 * ET) {
        const int e = tid & (ET - 1), which = tid / ET;
        float sc = part[(0 * 2 + which) * ET + e];
#pragma unroll
        for (int wv = 1; wv < 4; ++wv) sc += part[(wv * 2 + which) * ET + e];
        sc += fc2_b[0];
        if (e0 + e < B) prob[(size_t)which * B + e0 + e] = 1.f / (1.f + expf(-sc));
    }
}

// ---------------------------------------------------------------------------------------------------------
// metrics
// ---------------------------------------------------------------------------------------------------------
constexpr int MT_THREADS = 1024;
constexpr int MT_MAX = 16384;            // scores per call (2B): 128 KB of LDS keys

// float -> unsigned whose ascending order is the DESCENDING order of the floats (NaN-free scores)
__device__ __forceinline__ unsigned desc_key(float x)
{
    const unsigned u = __float_as_uint(x);
    const unsigned asc = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ~asc;
}

__global__ __launch_bounds__(MT_THREADS) void k_link_metrics(const float *__restrict__ pos, const float *__restrict__ neg, int B,
                                                             int n2, double *out, int accumulate)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64 *key = reinterpret_cast<u64 *>(smem);                             // [n2] (score key << 32) | label
    __shared__ double red[3][MT_THREADS / 64];
    __shared__ int part_t[MT_THREADS], part_s[MT_THREADS];
    const int tid = threadIdx.x, n = 2 * B;
    for (int i = tid; i < n2; i += MT_THREADS) {
        u64 kv = ~0ull;                                                   // padding sorts last
        if (i < n) kv = ((u64)desc_key(i < B ? pos[i] : neg[i - B]) << 32) | (i < B ? 1u : 0u);
        key[i] = kv;
    }
    __syncthreads();
    for (int k = 2; k <= n2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n2; i += MT_THREADS) {
                const int l = i ^ j;
                if (l > i) {
                    const u64 a = key[i], b = key[l];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { key[i] = b; key[l] = a; }
                }
            }
            __syncthreads();
        }
    // every thread owns a contiguous chunk: true positives (sum scan) and run starts (max scan)
    const int per = n2 / MT_THREADS > 0 ? n2 / MT_THREADS : 1;
    const int lo = tid * per, hi = lo + per < n2 ? lo + per : n2;
    int tsum = 0, smax = -1;
    for (int i = lo; i < hi && i < n; ++i) {
        tsum += (int)(key[i] & 1u);
        if (i == 0 || (unsigned)(key[i] >> 32) != (unsigned)(key[i - 1] >> 32)) smax = i;
    }
    part_t[tid] = tsum; part_s[tid] = smax;
    __syncthreads();
    if (tid == 0) {                                                       // 1024 partials: a serial pass is microseconds
        int t = 0, sm = -1;
        for (int q = 0; q < MT_THREADS; ++q) {
            const int a = part_t[q], b = part_s[q];
            part_t[q] = t; part_s[q] = sm;                                // exclusive prefixes
            t += a; if (b > sm) sm = b;
        }
    }
    __syncthreads();
    // second pass: at the LAST element of every run (a distinct threshold) add the curve terms.  tps / fps "before the
    // run" are those at index (run start - 1); since tps is needed at arbitrary earlier indices, every thread re-walks
    // its chunk carrying (tps, tps at the start of the current run)
    double ap = 0.0, auc = 0.0, acc = 0.0;
    {
        int t = part_t[tid], start = part_s[tid];
        // true positives before the run that is open at the chunk's first element: walk back is not possible, so the
        // prefix pass below carries it: tps_before_run = tps at (start - 1)
        // (computed by a third, tiny scan: the value of the sum scan at each run start)
        int t_before = 0;
        if (start >= 0) {
            // tps at start - 1 = number of labels in [0, start): chunk prefixes + a walk inside the chunk that holds `start`
            const int owner = start / per;
            int c = part_t[owner];
            for (int i = owner * per; i < start; ++i) c += (int)(key[i] & 1u);
            t_before = c;
        }
        const double np_ = (double)B, nn_ = (double)(n - B);
        for (int i = lo; i < hi && i < n; ++i) {
            if (i == 0 || (unsigned)(key[i] >> 32) != (unsigned)(key[i - 1] >> 32)) { start = i; t_before = t; }
            t += (int)(key[i] & 1u);
            const bool last = (i == n - 1) || (unsigned)(key[i + 1] >> 32) != (unsigned)(key[i] >> 32);
            if (last) {
                const double tps = t, fps = (double)(i + 1) - tps;
                const double tb = t_before, fb = (double)start - tb;
                ap += (tps / np_ - tb / np_) * (tps / (tps + fps));
                auc += (fps / nn_ - fb / nn_) * (tps / np_ + tb / np_) * 0.5;
            }
        }
    }
    for (int i = tid; i < B; i += MT_THREADS) acc += pos[i] >= neg[i] ? 1.0 : 0.0;
    // fixed-order reduction: lanes by DPP-free shuffles, waves through LDS, thread 0 last
    for (int d = 32; d > 0; d >>= 1) { ap += __shfl_down(ap, d); auc += __shfl_down(auc, d); acc += __shfl_down(acc, d); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = ap; red[1][tid >> 6] = auc; red[2][tid >> 6] = acc; }
    __syncthreads();
    if (tid == 0) {
        double a = 0.0, u = 0.0, c = 0.0;
        for (int q = 0; q < MT_THREADS / 64; ++q) { a += red[0][q]; u += red[1][q]; c += red[2][q]; }
        c /= (double)B;
        if (accumulate) { out[0] += a; out[1] += u; out[2] += c; }
        else { out[0] = a; out[1] = u; out[2] = c; }
    }
}

struct AffPlan { int Hp; size_t off_part, off_cnt, total; };
void aff_plan(int64_t max_B, int H, AffPlan &p)
{
    p.Hp = round_up16(H);
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 255) & ~(size_t)255; return r; };
    take(((size_t)2 * p.Hp * p.Hp + 2 * p.Hp) * 4);                       // packed weights first (offset 0)
    const size_t tiles = (size_t)((max_B + 15) / 16);
    p.off_cnt = take(tiles * 4);
    p.off_part = take((size_t)(p.Hp / 16) * 2 * tiles * 16 * 4);
    p.total = o;
}

}  // namespace

extern "C" int64_t zt_affinity_workspace_bytes(int64_t max_B, int32_t H)
{
    if (max_B <= 0 || H <= 0) return -1;
    if (H % 4 != 0 || (round_up16(H) != 208 && round_up16(H) != 304)) return -1;     // H = D (n_tppr + 1), D = 100, n_tppr in {1, 2}
    AffPlan p;
    aff_plan(max_B, H, p);
    return (int64_t)p.total;
}

extern "C" int zt_affinity(const float *emb_dev, int64_t B, int32_t H, const zt_affinity_weights *wt, float *prob_dev,
                           void *workspace_dev, int64_t ws_max_B, int32_t weights_ready, void *stream)
{
    if (!emb_dev || !wt || !prob_dev || !workspace_dev || B < 0 || H <= 0 || ws_max_B < B) {
        set_error("zt_affinity: bad argument");
        return ZT_ERR_ARG;
    }
    if (zt_affinity_workspace_bytes(ws_max_B > 0 ? ws_max_B : 1, H) < 0) {
        set_error("zt_affinity: H=%d unsupported (H = 200 or 300)", H);
        return ZT_ERR_UNSUPPORTED;
    }
    if (B == 0) return ZT_OK;
    AffPlan p;
    aff_plan(ws_max_B, H, p);
    hipStream_t s = (hipStream_t)stream;
    char *ws = reinterpret_cast<char *>(workspace_dev);
    float *packed = reinterpret_cast<float *>(ws);
    if (!weights_ready) {
        const int n = 2 * p.Hp * p.Hp + 2 * p.Hp;
        k_pack_affinity<<<(n + 255) / 256, 256, 0, s>>>(wt->fc1_w, wt->fc1_b, wt->fc2_w, H, p.Hp, packed);
        ZT_HIP(hipMemsetAsync(ws + p.off_cnt, 0, (size_t)((ws_max_B + 15) / 16) * 4, s));   // tile counters (self-resetting afterwards)
    }
    const long long tiles = (B + 15) / 16;
    // tiles x N-tiles waves; beyond ~3 waves per SIMD the waves stride over tiles with their weights in registers
    const dim3 grid((unsigned)(tiles < 160 ? tiles : 160), (unsigned)(p.Hp / 16));
    float *part = reinterpret_cast<float *>(ws + p.off_part);
    int *cnt = reinterpret_cast<int *>(ws + p.off_cnt);
    // small batches: the latency-organised kernel; large ones: the tiled one (both give the same probabilities to rounding;
    // where the switch sits: tools/exp/score_bench.py)
    constexpr long long tiled_min = 512;
    ZT_PROF_BEGIN(s, P_SCORE);
    if (B >= tiled_min) {
        // 16 edges per workgroup while that fills the chip once (two workgroups fit a CU), 32 beyond
        const int et = B <= 16 * 512 ? 16 : 32;
        const size_t lds = ((size_t)3 * et * (p.Hp + 4) + 4 * 2 * et) * 4;
        static size_t attr[4] = {0, 0, 0, 0};
        const int ki = (p.Hp == 304 ? 0 : 1) * 2 + (et == 16 ? 0 : 1);
        const void *fns[4] = {reinterpret_cast<const void *>(k_affinity_tiled<19, 16>), reinterpret_cast<const void *>(k_affinity_tiled<19, 32>),
                              reinterpret_cast<const void *>(k_affinity_tiled<13, 16>), reinterpret_cast<const void *>(k_affinity_tiled<13, 32>)};
        if (lds > attr[ki]) {
            ZT_HIP(hipFuncSetAttribute(fns[ki], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr[ki] = lds;
        }
        const unsigned wgs = (unsigned)((B + et - 1) / et);
        if (ki == 0)      k_affinity_tiled<19, 16><<<wgs, 256, lds, s>>>(emb_dev, B, H, packed, wt->fc2_b, prob_dev);
        else if (ki == 1) k_affinity_tiled<19, 32><<<wgs, 256, lds, s>>>(emb_dev, B, H, packed, wt->fc2_b, prob_dev);
        else if (ki == 2) k_affinity_tiled<13, 16><<<wgs, 256, lds, s>>>(emb_dev, B, H, packed, wt->fc2_b, prob_dev);
        else              k_affinity_tiled<13, 32><<<wgs, 256, lds, s>>>(emb_dev, B, H, packed, wt->fc2_b, prob_dev);
    } else if (p.Hp == 304) k_affinity<19><<<grid, 64, 0, s>>>(emb_dev, B, H, packed, wt->fc2_b, part, cnt, prob_dev);
    else                    k_affinity<13><<<grid, 64, 0, s>>>(emb_dev, B, H, packed, wt->fc2_b, part, cnt, prob_dev);
    ZT_PROF_END(s, P_SCORE);
    ZT_LAUNCH_CHECK();
    return ZT_OK;
}

extern "C" int zt_link_metrics(const float *pos_dev, const float *neg_dev, int64_t B, double *out_dev, int32_t accumulate,
                               void *stream)
{
    if (!pos_dev || !neg_dev || !out_dev || B <= 0) { set_error("zt_link_metrics: bad argument"); return ZT_ERR_ARG; }
    if (2 * B > MT_MAX) { set_error("zt_link_metrics: %lld scores per call, at most %d", (long long)(2 * B), MT_MAX); return ZT_ERR_UNSUPPORTED; }
    int n2 = MT_THREADS;                                                  // >= one element per thread keeps the chunks simple
    while (n2 < 2 * B) n2 <<= 1;
    const size_t lds = (size_t)n2 * 8;
    static size_t attr = 0;
    if (lds > 48 * 1024 && lds > attr) {
        ZT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_link_metrics), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = lds;
    }
    k_link_metrics<<<1, MT_THREADS, lds, (hipStream_t)stream>>>(pos_dev, neg_dev, (int)B, n2, out_dev, accumulate);
    ZT_LAUNCH_CHECK();
    return ZT_OK;
}
