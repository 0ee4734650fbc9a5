// Micro-benchmark: LDS read rate of an 8-byte-per-lane access on gfx950 as ds_read2_b32 (two dwords,
// offsets 0 and 1), ds_read_b64 at an 8-byte aligned address and ds_read_b64 at an address that is
// only 4-byte aligned.  Prints cycles per wave-instruction per CU and checks the values returned.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__device__ __forceinline__ f2 rd(unsigned addr)
{
    f2 v;
    if (MODE == 0) asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(v) : "v"(addr));
    else asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}

// STRIDE: dwords between consecutive lanes
template <int MODE, int ODD, int STRIDE>
__global__ void __launch_bounds__(256) k(float* out, int iters)
{
    __shared__ float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned addr = (unsigned)(size_t)lds + 4u * (unsigned)(((lane * STRIDE) % 2048) + wave * 2048 + ODD);
    f2 acc = {0.f, 0.f};
    for (int i = 0; i < iters; ++i) {
        f2 a = rd<MODE>(addr), b = rd<MODE>(addr + 4096 * 4), c = rd<MODE>(addr + 8), d = rd<MODE>(addr + 4096 * 4 + 8);
        f2 e = rd<MODE>(addr + 16), f = rd<MODE>(addr + 4096 * 4 + 16), g = rd<MODE>(addr + 24), h = rd<MODE>(addr + 4096 * 4 + 24);
        // the registers are operands of the wait, so that no use of them can be scheduled above it
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)::"memory");
        acc += a + b + c + d + e + f + g + h;
        asm volatile("" : "+v"(addr));
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc.x + 0.5f * acc.y;
}

template <int MODE, int ODD, int STRIDE> void run(const char* name, float* d, float* h)
{
    const int iters = 4000, blocks = 256 * 4;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, ODD, STRIDE>), dim3(blocks), dim3(256), 0, 0, d, 10);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, ODD, STRIDE>), dim3(blocks), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(h, d, 256 * sizeof(float), hipMemcpyDeviceToHost);
    // expected value of thread 0..255 of block 0
    int bad = 0;
    for (int t = 0; t < 256; ++t) {
        int lane = t & 63, wave = t >> 6;
        int base = ((lane * STRIDE) % 2048) + wave * 2048 + ODD;
        double x = 0, y = 0;
        for (int q = 0; q < 4; ++q) { x += base + 2 * q; y += base + 2 * q + 1; x += base + 4096 + 2 * q; y += base + 4096 + 2 * q + 1; }
        float want = (float)(iters * x) + 0.5f * (float)(iters * y);
        if (fabsf(h[t] - want) > 1e-3f * fabsf(want)) ++bad;
    }
    double instr_per_cu = (double)blocks / 256.0 * 4 /*waves*/ * iters * 8;
    double cyc = ms * 1e-3 * 2.4e9 / instr_per_cu;
    printf("%-40s %.3f ms  %.2f cyc per wave-instruction per CU (@2.4GHz) = %.0f B/clk/CU  %s\n", name, ms, cyc, 512.0 / cyc,
           bad ? "VALUES WRONG" : "values ok");
}

int main()
{
    float *d, *h = (float*)malloc(256 * sizeof(float));
    (void)hipMalloc(&d, 256 * 4 * 256 * sizeof(float));
    run<0, 0, 2>("ds_read2_b32 stride 2, even", d, h);
    run<0, 1, 2>("ds_read2_b32 stride 2, odd", d, h);
    run<1, 0, 2>("ds_read_b64  stride 2, aligned", d, h);
    run<1, 1, 2>("ds_read_b64  stride 2, dword-aligned", d, h);
    run<0, 0, 117>("ds_read2_b32 stride 117, even", d, h);
    run<1, 0, 118>("ds_read_b64  stride 118, aligned", d, h);
    run<1, 1, 118>("ds_read_b64  stride 118, dword-aligned", d, h);
    run<1, 1, 117>("ds_read_b64  stride 117, mixed alignment", d, h);
    return 0;
}
