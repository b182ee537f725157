// Micro-benchmark: cost of the rank pass of a hub hop for ONE lone wave (cycles per call), and variants.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
#define WAVE 64
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ double readlane_f64(double x, int src)
{
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)(unsigned)(b & 0xffffffffll), src);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
// A: the product's rank pass (f64 compare against an SGPR pair)
__device__ __forceinline__ int rank_a(double vc, u64 live)
{
    int l0 = 0, l1 = 0, l2 = 0, l3 = 0;
#pragma unroll
    for (int q0 = 0; q0 < WAVE; q0 += 8) {
        if (((live >> q0) & 0xffull) == 0ull) continue;
        l0 += (readlane_f64(vc, q0 + 0) < vc) ? 1 : 0;
        l1 += (readlane_f64(vc, q0 + 1) < vc) ? 1 : 0;
        l2 += (readlane_f64(vc, q0 + 2) < vc) ? 1 : 0;
        l3 += (readlane_f64(vc, q0 + 3) < vc) ? 1 : 0;
        l0 += (readlane_f64(vc, q0 + 4) < vc) ? 1 : 0;
        l1 += (readlane_f64(vc, q0 + 5) < vc) ? 1 : 0;
        l2 += (readlane_f64(vc, q0 + 6) < vc) ? 1 : 0;
        l3 += (readlane_f64(vc, q0 + 7) < vc) ? 1 : 0;
    }
    return (l0 + l1) + (l2 + l3);
}
// B: unsigned 64-bit compare of the bit patterns (weights are >= 0: same order)
__device__ __forceinline__ int rank_b(double vc, u64 live)
{
    const u64 b = (u64)__double_as_longlong(vc);
    int l0 = 0, l1 = 0, l2 = 0, l3 = 0;
#pragma unroll
    for (int q0 = 0; q0 < WAVE; q0 += 8) {
        if (((live >> q0) & 0xffull) == 0ull) continue;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, q0 + t);
            const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), q0 + t);
            const u64 o = ((u64)hi << 32) | lo;
            const int c = o < b ? 1 : 0;
            if ((t & 3) == 0) l0 += c; else if ((t & 3) == 1) l1 += c; else if ((t & 3) == 2) l2 += c; else l3 += c;
        }
    }
    return (l0 + l1) + (l2 + l3);
}
// C: values through LDS, read back as broadcasts (b128 = two values per read)
__device__ __forceinline__ int rank_c(double vc, u64 live, double *buf)
{
    buf[lane_id()] = vc;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int l0 = 0, l1 = 0, l2 = 0, l3 = 0;
    typedef double d2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int q0 = 0; q0 < WAVE; q0 += 8) {
        if (((live >> q0) & 0xffull) == 0ull) continue;
        const d2 a = *reinterpret_cast<const d2 *>(buf + q0), b = *reinterpret_cast<const d2 *>(buf + q0 + 2),
                 c = *reinterpret_cast<const d2 *>(buf + q0 + 4), d = *reinterpret_cast<const d2 *>(buf + q0 + 6);
        l0 += (a[0] < vc) ? 1 : 0; l1 += (a[1] < vc) ? 1 : 0; l2 += (b[0] < vc) ? 1 : 0; l3 += (b[1] < vc) ? 1 : 0;
        l0 += (c[0] < vc) ? 1 : 0; l1 += (c[1] < vc) ? 1 : 0; l2 += (d[0] < vc) ? 1 : 0; l3 += (d[1] < vc) ? 1 : 0;
    }
    __builtin_amdgcn_wave_barrier();
    return (l0 + l1) + (l2 + l3);
}
// mate: 0 none; 1 = waves 4-7 (the SIMD mates of waves 0-3) run a float64 FMA loop at equal priority;
// 2 = the mates run it at priority 0 while waves 0-3 are at priority 3
__global__ void k2(int mate, int iters, double *out, long long *cyc)
{
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const u64 live = ((1ull << 20) - 1ull) | (((1ull << 21) - 1ull) << 32);
    if (wave >= 4) {
        if (mate == 0) return;
        if (mate == 2) __builtin_amdgcn_s_setprio(0);
        double a = lane * 1e-3, b = 1.0000001, c = 0.5;
        for (int it = 0; it < iters * 60; ++it) { a = a * b + c; b = b * 0.9999999 + 1e-9; c = c * b + a; }
        out[threadIdx.x] = a + b + c;
        return;
    }
    if (mate == 2) __builtin_amdgcn_s_setprio(3);
    double v = (double)((lane * 37) % 64) * 0.125 + 1.0;
    int acc = 0;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const int r = rank_a(v, live);
        acc += r;
        v += (double)(r & 1) * 1e-9;
    }
    const long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = v + acc;
    if (threadIdx.x == 0) cyc[3 + mate] = t1 - t0;
}
__global__ void k(int mode, int iters, double *out, long long *cyc)
{
    __shared__ double buf[64];
    const int lane = lane_id();
    const u64 live = ((1ull << 20) - 1ull) | (((1ull << 21) - 1ull) << 32);
    double v = (double)((lane * 37) % 64) * 0.125 + 1.0;
    int acc = 0;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        int r;
        if (mode == 0) r = rank_a(v, live);
        else if (mode == 1) r = rank_b(v, live);
        else r = rank_c(v, live, buf);
        acc += r;
        v += (double)(r & 1) * 1e-9;           // dependency between iterations, as on the chain
    }
    const long long t1 = __builtin_readcyclecounter();
    out[lane] = v + acc;
    if (lane == 0) cyc[mode] = t1 - t0;
}
int main()
{
    double *out; long long *cyc, h[6];
    hipMalloc(&out, 512 * 8); hipMalloc(&cyc, 6 * 8);
    const int iters = 2000;
    for (int m = 0; m < 3; ++m) { k<<<1, 64>>>(m, 10, out, cyc); hipDeviceSynchronize(); k<<<1, 64>>>(m, iters, out, cyc); hipDeviceSynchronize(); }
    for (int m = 0; m < 3; ++m) { k2<<<1, 512>>>(m, 10, out, cyc); hipDeviceSynchronize(); k2<<<1, 512>>>(m, iters, out, cyc); hipDeviceSynchronize(); }
    hipMemcpy(h, cyc, 48, hipMemcpyDeviceToHost);
    const char *names[3] = {"f64 compare vs readlane pair", "u64 compare vs readlane pair", "LDS broadcast b128"};
    for (int m = 0; m < 3; ++m) printf("%-32s %8.1f ticks / rank pass (s_memtime ticks; 48 comparisons)\n", names[m], (double)h[m] / iters);
    const char *n2[3] = {"8-wave WG, mates idle", "mates busy, equal priority", "mates busy at prio 0, ranker at prio 3"};
    for (int m = 0; m < 3; ++m) printf("%-40s %8.1f cycles / rank pass\n", n2[m], (double)h[3 + m] / iters);
    return 0;
}
