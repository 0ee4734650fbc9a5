// Micro-benchmark: cost of the 8-tap trilinear gather pattern through different load forms (L1 resident).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(4))) pf2 { float x, y; };
constexpr int P = 80, Q = 80 * 12;   // row pitch, plane stride (floats) of a small L1-resident brick

template <int MODE>
__global__ void __launch_bounds__(256) k(const float* __restrict__ src, float* out, int iters)
{
    __shared__ float lds[Q * 6];
    for (int q = threadIdx.x; q < Q * 6; q += 256) lds[q] = src[q];
    __syncthreads();
    int lane = threadIdx.x & 63, row = threadIdx.x >> 6;
    float acc = 0.f;
    int base = 0;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, Q * 6 * 4, 0x00020000);
    for (int i = 0; i < iters; ++i) {
        base = (base * 5 + 3) & 7;   // wave-uniform shift, changes every iteration
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            int o = lane + base + (row + u) * P + u * Q;   // lower corner of this lane's cell
            if (MODE == 0) {        // 4 x global dwordx2 (4-byte aligned)
                pf2 a = *reinterpret_cast<const pf2*>(src + o), b = *reinterpret_cast<const pf2*>(src + o + P);
                pf2 c = *reinterpret_cast<const pf2*>(src + o + Q), d = *reinterpret_cast<const pf2*>(src + o + Q + P);
                acc += a.x + a.y + b.x + b.y + c.x + c.y + d.x + d.y;
            } else if (MODE == 1) { // 8 x global dword
                acc += src[o] + src[o + 1] + src[o + P] + src[o + P + 1] + src[o + Q] + src[o + Q + 1] + src[o + Q + P] + src[o + Q + P + 1];
            } else if (MODE == 2) { // 8 x buffer dword, one voffset
                int vo = o * 4;
                acc += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo, 0, 0)) +
                       __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo + 4, 0, 0)) +
                       __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo, P * 4, 0)) +
                       __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo + 4, P * 4, 0)) +
                       __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo, Q * 4, 0)) +
                       __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo + 4, Q * 4, 0)) +
                       __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo, (Q + P) * 4, 0)) +
                       __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo + 4, (Q + P) * 4, 0));
            } else if (MODE == 3) { // 4 x lds read2_b32
                acc += lds[o] + lds[o + 1] + lds[o + P] + lds[o + P + 1] + lds[o + Q] + lds[o + Q + 1] + lds[o + Q + P] + lds[o + Q + P + 1];
            } else if (MODE == 4) { // 4 x global dword (half the taps) to see per-instruction cost
                acc += src[o] + src[o + P] + src[o + Q] + src[o + Q + P];
            }
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
    double gathers = (double)blocks * 4 * iters * 3;      // 8-tap gathers per wave
    double cyc = ms * 1e-3 * 2.3e9 * 256 / gathers;
    printf("%-34s %.3f ms  %.1f cyc per wave-gather per CU\n", name, ms, cyc);
}
int main()
{
    float *s, *d; (void)hipMalloc(&s, 1 << 20); (void)hipMemset(s, 0, 1 << 20); (void)hipMalloc(&d, 4);
    run<0>("4 x global dwordx2", s, d);
    run<1>("8 x global dword", s, d);
    run<2>("8 x buffer dword (1 voffset)", s, d);
    run<3>("lds 8 taps", s, d);
    run<4>("4 x global dword (half)", s, d);
    return 0;
}
