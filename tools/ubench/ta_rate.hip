// Micro-benchmark: L1-hit vector-memory and LDS gather issue rates on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) pf2 { float x, y; };

template <int MODE>
__global__ void __launch_bounds__(256) k(const float* __restrict__ src, float* out, int iters)
{
    __shared__ float lds[4096];
    for (int q = threadIdx.x; q < 4096; q += 256) lds[q] = src[q];
    __syncthreads();
    int lane = threadIdx.x;
    float acc = 0.f;
    int idx = lane;           // per-lane element index, small footprint -> L1 resident
    int base = 0;
    for (int i = 0; i < iters; ++i) {
        base = (base * 5 + 17) & 1023;   // wave-uniform, changes every iteration: nothing hoists
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            int o = (idx + u * 264 + base) & 2047;
            if (MODE == 0) { acc += src[o]; }
            else if (MODE == 1) { pf2 v = *reinterpret_cast<const pf2*>(src + o); acc += v.x + v.y; }          // 4B stride, 8B load
            else if (MODE == 2) { f2 v = *reinterpret_cast<const f2*>(src + 2 * (o & 1023)); acc += v.x + v.y; } // 8B stride
            else if (MODE == 3) { f4 v = *reinterpret_cast<const f4*>(src + 4 * (o & 511)); acc += v.x + v.w; }   // 16B stride
            else if (MODE == 4) { acc += lds[o]; }
            else if (MODE == 5) { acc += lds[o] + lds[o + 1]; }                                                   // ds_read2_b32
            else if (MODE == 6) { f2 v = *reinterpret_cast<const f2*>(lds + 2 * (o & 1023)); acc += v.x + v.y; }  // ds_read_b64
            else if (MODE == 7) { acc += src[o] + src[o + 1]; }                                                   // two dword loads
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}
template <int MODE> void run(const char* name, const float* s, float* d)
{
    int iters = 2000, blocks = 256 * 8;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, s, d, 10);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, s, d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double wave_instr = (double)blocks * 4 * iters * 8;   // load groups per wave
    double cyc = ms * 1e-3 * 2.3e9 * 256 / wave_instr;    // cycles per group per CU at 2.3 GHz
    printf("%-28s %.3f ms  %.1f cyc per wave-group per CU\n", name, ms, cyc);
}
int main()
{
    float *s, *d; (void)hipMalloc(&s, 1 << 20); (void)hipMemset(s, 0, 1 << 20); (void)hipMalloc(&d, 4);
    run<0>("global dword (4B stride)", s, d);
    run<1>("global dwordx2 4B-stride", s, d);
    run<7>("global 2 x dword (o, o+1)", s, d);
    run<2>("global dwordx2 8B-stride", s, d);
    run<3>("global dwordx4 16B-stride", s, d);
    run<4>("lds b32", s, d);
    run<5>("lds 2 x b32 (o, o+1)", s, d);
    run<6>("lds b64 aligned", s, d);
    return 0;
}
