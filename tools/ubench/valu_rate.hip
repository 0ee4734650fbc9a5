// Micro-benchmark: issue rate of plain vs packed fp32 VALU ops on gfx950 at full occupancy.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#pragma clang fp contract(off)
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed)
{
    float a0 = threadIdx.x * 1e-3f + seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const float c = 1.0000001f; const f2 c2 = {c, c};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {          // 8 scalar mul
            a0 *= c; a1 *= c; a2 *= c; a3 *= c; a4 *= c; a5 *= c; a6 *= c; a7 *= c;
        } else if (MODE == 1) {   // 4 packed mul (same 8 flops)
            p0 *= c2; p1 *= c2; p2 *= c2; p3 *= c2;
        } else if (MODE == 2) {   // 8 scalar add
            a0 += c; a1 += c; a2 += c; a3 += c; a4 += c; a5 += c; a6 += c; a7 += c;
        } else if (MODE == 3) {   // 4 packed add
            p0 += c2; p1 += c2; p2 += c2; p3 += c2;
        } else if (MODE == 4) {   // 8 scalar fma
            a0 = __builtin_fmaf(a0, c, c); a1 = __builtin_fmaf(a1, c, c); a2 = __builtin_fmaf(a2, c, c); a3 = __builtin_fmaf(a3, c, c);
            a4 = __builtin_fmaf(a4, c, c); a5 = __builtin_fmaf(a5, c, c); a6 = __builtin_fmaf(a6, c, c); a7 = __builtin_fmaf(a7, c, c);
        } else if (MODE == 5) {   // 4 packed fma
            p0 = __builtin_elementwise_fma(p0, c2, c2); p1 = __builtin_elementwise_fma(p1, c2, c2);
            p2 = __builtin_elementwise_fma(p2, c2, c2); p3 = __builtin_elementwise_fma(p3, c2, c2);
        }
    }
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if (r == 12345.678f) out[0] = r;
}
template <int MODE> void run(const char* name, float* d)
{
    int iters = 20000, blocks = 256 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 100, 1.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double lane_ops = (double)blocks * 256 * iters * 8;   // flops-lanes (8 values updated per iter)
    double waves = (double)blocks * 4;
    double instr = waves * iters * ((MODE & 1) ? 4 : 8);
    // cycles per wave-instruction per SIMD assuming 2.4 GHz, 1024 SIMDs
    double cyc = ms * 1e-3 * 2.4e9 * 1024 / instr;
    printf("%-12s %.3f ms  %.2f Tlane-op/s  ~%.2f cyc/instr/SIMD(@2.4GHz)\n", name, ms, lane_ops / ms / 1e9, cyc);
}
int main()
{
    float* d; hipMalloc(&d, 4);
    run<0>("mul", d); run<1>("pk_mul", d); run<2>("add", d); run<3>("pk_add", d); run<4>("fma", d); run<5>("pk_fma", d);
    return 0;
}
