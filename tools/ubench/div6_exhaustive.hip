// Exhaustive check of the binary32-only form of x / 6 over all 2^32 bit patterns, four values at a
// time as the Jacobi kernels use it:
//     q0 = x * r;  e = fma(-6, q0, x);  q = fma(e, r, q0)          (r = RN32(1/6))
// is the correctly rounded quotient whenever q0 is a normal number or +0 (one Newton correction from
// a quotient within an ulp, the residual is exact, and x / 6 cannot sit on a rounding boundary
// because 6 * boundary is odd in units x is an even multiple of).  Where some q0 of the four is
// subnormal, infinite, NaN or -0 (ties exist on the subnormal grid, inf - inf, the sign of zero), the
// four are redone as RN32(RN64(x) * RN64(1/6)).  Compared bit for bit with the IEEE division.
// Measured on MI355X: the guarded form differs on 0 of 2^32 patterns (the fast path alone on 2 796 205),
// and costs 15.5 cycles per (add + division) against 16.2 for the binary64 form in use: four
// instructions and a branch in place of three, 4 % -- not worth a second code path in the kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang fp contract(off)
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float d_f64(float x) { return (float)((double)x * 0x1.5555555555555p-3); }
__device__ __forceinline__ f4 div6x4(f4 x)
{
    const float r = 0x1.555556p-3f;
    const f4 q0 = x * r;
    constexpr int kBad = 0x2B7;     // sNaN qNaN -inf -subnormal -0 +subnormal +inf
    const bool bad = __builtin_amdgcn_classf(q0.x, kBad) | __builtin_amdgcn_classf(q0.y, kBad) |
                     __builtin_amdgcn_classf(q0.z, kBad) | __builtin_amdgcn_classf(q0.w, kBad);
    f4 q;
    q.x = __builtin_fmaf(__builtin_fmaf(-6.0f, q0.x, x.x), r, q0.x);
    q.y = __builtin_fmaf(__builtin_fmaf(-6.0f, q0.y, x.y), r, q0.y);
    q.z = __builtin_fmaf(__builtin_fmaf(-6.0f, q0.z, x.z), r, q0.z);
    q.w = __builtin_fmaf(__builtin_fmaf(-6.0f, q0.w, x.w), r, q0.w);
    if (__builtin_amdgcn_ballot_w64(bad) != 0) q = f4{d_f64(x.x), d_f64(x.y), d_f64(x.z), d_f64(x.w)};
    return q;
}
// GUARD = 1: the form above; 0: the fast path alone (counts what the guard is for)
template <int GUARD>
__global__ void __launch_bounds__(256) k_check(unsigned long long* mism, unsigned* first)
{
    const unsigned long long t = (unsigned long long)blockIdx.x * 256 + threadIdx.x;   // 2^30 threads
    unsigned bits[4];
    f4 x;
    for (int c = 0; c < 4; ++c) {
        // lane-neighbouring patterns in one call, and a stride that mixes binades inside a wave
        bits[c] = (unsigned)(t * 4 + c) * 2654435761u;     // odd multiplier: a bijection of 2^32
        ((float*)&x)[c] = __uint_as_float(bits[c]);
    }
    f4 q;
    if (GUARD) q = div6x4(x);
    else {
        const float r = 0x1.555556p-3f;
        for (int c = 0; c < 4; ++c) {
            float xc = ((float*)&x)[c], q0 = xc * r;
            ((float*)&q)[c] = __builtin_fmaf(__builtin_fmaf(-6.0f, q0, xc), r, q0);
        }
    }
    unsigned bad = 0;
    for (int c = 0; c < 4; ++c) {
        const float xc = ((float*)&x)[c];
        const float want = xc / 6.0f;
        const float got = ((float*)&q)[c];
        const bool same = __float_as_uint(want) == __float_as_uint(got) || (want != want && got != got);
        if (!same) {
            ++bad;
            atomicMin(first, bits[c]);
        }
    }
    if (bad) atomicAdd(mism, (unsigned long long)bad);
}
template <int MODE>
__global__ void __launch_bounds__(256) k_rate(float* out, int iters, float seed)
{
    f4 a[2];
    for (int j = 0; j < 2; ++j) a[j] = f4{threadIdx.x * 1e-3f + seed + j, seed + 1.f, seed + 2.f + j, seed + 3.f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f4 x = a[j] + 7.0f;
            a[j] = MODE == 0 ? f4{d_f64(x.x), d_f64(x.y), d_f64(x.z), d_f64(x.w)} : div6x4(x);
        }
    }
    float r = a[0].x + a[0].y + a[0].z + a[0].w + a[1].x + a[1].y + a[1].z + a[1].w;
    if (r == 12345.678f) out[0] = r;
}
template <int MODE> void rate(const char* name, float* d)
{
    int iters = 10000, blocks = 256 * 8;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, d, 100, 1.f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double waves = (double)blocks * 4, divs = waves * iters * 8;
    printf("%-8s %.3f ms  ~%.2f cycles per (add + division) per wave and SIMD (@2.4 GHz)\n", name, ms,
           ms * 1e-3 * 2.4e9 * 1024 / divs);
}
int main()
{
    unsigned long long* mism; unsigned* first;
    (void)hipMalloc(&mism, 8); (void)hipMalloc(&first, 4);
    for (int guard = 1; guard >= 0; --guard) {
        (void)hipMemset(mism, 0, 8); (void)hipMemset(first, 0xff, 4);
        if (guard) hipLaunchKernelGGL(k_check<1>, dim3(1u << 22), dim3(256), 0, 0, mism, first);
        else hipLaunchKernelGGL(k_check<0>, dim3(1u << 22), dim3(256), 0, 0, mism, first);
        unsigned long long m = 0; unsigned f = 0;
        (void)hipMemcpy(&m, mism, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&f, first, 4, hipMemcpyDeviceToHost);
        printf("%s: %llu of 4294967296 bit patterns differ from x / 6.0f", guard ? "guarded form" : "fast path alone", m);
        if (m) printf(" (smallest pattern 0x%08x)", f);
        printf("\n");
    }
    float* d; (void)hipMalloc(&d, 4);
    rate<0>("f64", d); rate<1>("guarded", d);
    return 0;
}
