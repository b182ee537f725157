// Probe: can a stream wait (command-processor side, no kernel) for a word of plain device memory that a RUNNING kernel of
// another stream writes?  hipStreamWaitValue32 on hipMalloc memory, on hipMallocSignalMemory memory, and on CU-masked
// streams; prints what is supported and how long a satisfied / unsatisfied wait holds the stream up.
//   hipcc --offload-arch=gfx950 -O2 tools/exp/waitvalue_probe.hip -o tools/out/waitvalue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_writer(int *word, int value, long long ticks)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    __hip_atomic_store(word, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void k_stamp(long long *out) { if (threadIdx.x == 0) *out = wall_clock64(); }

static int run(const char *what, int *word, hipStream_t a, hipStream_t b)
{
    long long *st;
    CK(hipMalloc(&st, 4 * sizeof(long long)));
    CK(hipMemset(word, 0, 8));
    CK(hipDeviceSynchronize());
    // unsatisfied at first: the writer sleeps 200 us (20000 ticks of 100 MHz)
    k_stamp<<<1, 64, 0, b>>>(st);
    k_writer<<<1, 64, 0, a>>>(word, 7, 20000);
    hipError_t e = hipStreamWaitValue32(b, word, 7, hipStreamWaitValueGte, 0xffffffffu);
    if (e != hipSuccess) { printf("%-28s hipStreamWaitValue32 -> %s\n", what, hipGetErrorString(e)); (void)hipGetLastError(); (void)hipDeviceSynchronize(); return 0; }
    k_stamp<<<1, 64, 0, b>>>(st + 1);
    CK(hipDeviceSynchronize());
    // satisfied on arrival
    k_stamp<<<1, 64, 0, b>>>(st + 2);
    CK(hipStreamWaitValue32(b, word, 7, hipStreamWaitValueGte, 0xffffffffu));
    k_stamp<<<1, 64, 0, b>>>(st + 3);
    CK(hipDeviceSynchronize());
    long long h[4];
    CK(hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost));
    printf("%-28s waited %.1f us for a word written 200 us after launch; a satisfied wait between two kernels: %.1f us\n", what,
           (h[1] - h[0]) / 100.0, (h[3] - h[2]) / 100.0);
    // the same pair of stamps with nothing in between, for reference
    k_stamp<<<1, 64, 0, b>>>(st + 2);
    k_stamp<<<1, 64, 0, b>>>(st + 3);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost));
    printf("%-28s two kernels back to back: %.1f us\n", what, (h[3] - h[2]) / 100.0);
    (void)hipFree(st);
    return 0;
}

int main()
{
    int can = -1;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    int *plain, *sig;
    CK(hipMalloc(&plain, 64));
    if (run("hipMalloc, plain streams", plain, a, b)) return 1;
    if (hipExtMallocWithFlags((void **)&sig, 8, hipMallocSignalMemory) == hipSuccess) {
        if (run("signal memory, plain streams", sig, a, b)) return 1;
    } else { printf("hipMallocSignalMemory: not available\n"); (void)hipGetLastError(); sig = nullptr; }
    // CU-masked streams: 64 CUs / the other 192
    std::vector<uint32_t> m1(8, 0), m2(8, 0);
    for (int c = 0; c < 256; ++c) (c < 64 ? m1 : m2)[c / 32] |= 1u << (c % 32);
    hipStream_t ma, mb;
    CK(hipExtStreamCreateWithCUMask(&ma, 8, m1.data()));
    CK(hipExtStreamCreateWithCUMask(&mb, 8, m2.data()));
    if (run("hipMalloc, masked streams", plain, ma, mb)) return 1;
    if (sig && run("signal memory, masked streams", sig, ma, mb)) return 1;
    return 0;
}
