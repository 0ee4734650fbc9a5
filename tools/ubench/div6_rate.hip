// Micro-benchmark: cost of x / 6 in three forms on gfx950 at full occupancy (8 independent values per
// thread, so the issue rate is measured, not the latency):
//   f64    RN32(RN64(x) * RN64(1/6))            v_cvt_f64_f32, v_mul_f64, v_cvt_f32_f64
//   fma    q = x * r; q + fma(-6, q, x) * r      v_mul_f32, v_fma_f32, v_fma_f32   (r = RN32(1/6))
//   div    x / 6.f                               the IEEE division sequence
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang fp contract(off)
__device__ __forceinline__ float d_f64(float x) { return (float)((double)x * 0x1.5555555555555p-3); }
__device__ __forceinline__ float d_fma(float x)
{
    const float r = 0x1.555556p-3f;
    const float q = x * r;
    return __builtin_fmaf(__builtin_fmaf(-6.0f, q, x), r, q);
}
__device__ __forceinline__ float d_div(float x) { return x / 6.0f; }
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed)
{
    float a[8];
    for (int j = 0; j < 8; ++j) a[j] = threadIdx.x * 1e-3f + seed + j;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float x = a[j] + 7.0f;      // keeps the values in range: one add per division in every mode
            a[j] = MODE == 0 ? d_f64(x) : (MODE == 1 ? d_fma(x) : d_div(x));
        }
    }
    float r = 0.f;
    for (int j = 0; j < 8; ++j) r += a[j];
    if (r == 12345.678f) out[0] = r;
}
template <int MODE> void run(const char* name, float* d)
{
    int iters = 10000, blocks = 256 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 100, 1.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double waves = (double)blocks * 4, divs = waves * iters * 8;
    printf("%-6s %.3f ms  ~%.2f cycles per (add + division) per wave and SIMD (@2.4 GHz)\n", name, ms,
           ms * 1e-3 * 2.4e9 * 1024 / divs);
}
int main()
{
    float* d; hipMalloc(&d, 4);
    run<0>("f64", d); run<1>("fma", d); run<2>("div", d);
    // exhaustive agreement of the two short forms with the true quotient is checked by the test suite
    return 0;
}
