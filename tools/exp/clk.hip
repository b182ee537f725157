// What does the shader clock do?  s_memtime (core cycles) against s_memrealtime (100 MHz) for one busy wave,
// (a) alone on the chip, (b) while an MFMA kernel keeps the other CUs busy.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(long long *o, int iters)
{
    double a = threadIdx.x * 1e-3, b = 1.0000001;
    const long long c0 = __builtin_readcyclecounter(), r0 = (long long)wall_clock64();
    for (int i = 0; i < iters; ++i) { a = a * b + 0.5; }
    const long long c1 = __builtin_readcyclecounter(), r1 = (long long)wall_clock64();
    if (threadIdx.x == 0) { o[0] = c1 - c0; o[1] = r1 - r0; }
    if (a == 123.0) o[2] = 1;
}
__global__ __launch_bounds__(256) void burn(float *out, int iters)
{
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main()
{
    long long *o, h[2]; float *f;
    hipMalloc(&o, 64); hipMalloc(&f, 1024 * 256 * 4);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    for (int rep = 0; rep < 2; ++rep) {
        probe<<<1, 64, 0, s1>>>(o, 200000); hipDeviceSynchronize();
        hipMemcpy(h, o, 16, hipMemcpyDeviceToHost);
        printf("alone:           %lld core cycles in %lld ticks of 10 ns -> %.0f MHz\n", h[0], h[1], h[0] / (h[1] * 0.01));
    }
    burn<<<1000, 256, 0, s2>>>(f, 400000);
    probe<<<1, 64, 0, s1>>>(o, 200000);
    hipDeviceSynchronize();
    hipMemcpy(h, o, 16, hipMemcpyDeviceToHost);
    printf("beside MFMA load: %lld core cycles in %lld ticks of 10 ns -> %.0f MHz\n", h[0], h[1], h[0] / (h[1] * 0.01));
    return 0;
}
