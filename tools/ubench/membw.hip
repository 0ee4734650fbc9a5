// Achievable HBM bandwidth on one MI355X: read-only, write-only, copy (1 read : 1 write) and
// the Jacobi mix (2 reads : 1 write), 16 B per lane, with full-line and with 224-of-256-byte rows.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k_read(const f4* __restrict__ a, f4* out, size_t n)
{
    f4 acc = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += a[i];
    if (acc.x == 12345.f) out[0] = acc;
}
__global__ void k_write(f4* __restrict__ a, size_t n, int partial)
{
    f4 v = {1, 2, 3, 4};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (!partial || ((i & 15) != 0 && (i & 15) != 15)) a[i] = v;
}
__global__ void k_copy(const f4* __restrict__ a, f4* __restrict__ b, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void k_mix21(const f4* __restrict__ a, const f4* __restrict__ c, f4* __restrict__ b, size_t n, int partial)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        f4 v = a[i] + c[i];
        if (!partial || ((i & 15) != 0 && (i & 15) != 15)) b[i] = v;
    }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main(int argc, char** argv)
{
    size_t mib = argc > 1 ? atoi(argv[1]) : 1024;
    size_t bytes = mib << 20, n = bytes / 16;
    const double gib = bytes / 1e9;   // GB per buffer
    printf("buffers of %zu MiB\n", mib);
    f4 *a, *b, *c;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&c, bytes));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes)); CK(hipMemset(c, 0, bytes));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int grid = 256 * 8, block = 512;
    auto timeit = [&](const char* name, double gb, auto launch) {
        for (int i = 0; i < 2; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %7.1f us  %6.2f TB/s\n", name, ms * 100, gb * 10 / ms / 1e3 * 1e3);
    };
    timeit("read", gib, [&] { hipLaunchKernelGGL(k_read, grid, block, 0, 0, a, b, n); });
    timeit("write", gib, [&] { hipLaunchKernelGGL(k_write, grid, block, 0, 0, b, n, 0); });
    timeit("write, 224/256 B rows", gib * 14 / 16, [&] { hipLaunchKernelGGL(k_write, grid, block, 0, 0, b, n, 1); });
    timeit("copy 1 read + 1 write", 2 * gib, [&] { hipLaunchKernelGGL(k_copy, grid, block, 0, 0, a, b, n); });
    timeit("mix 2 reads + 1 write", 3 * gib, [&] { hipLaunchKernelGGL(k_mix21, grid, block, 0, 0, a, c, b, n, 0); });
    timeit("mix 2r + 1w, 224/256 rows", (2 + 14. / 16) * gib, [&] { hipLaunchKernelGGL(k_mix21, grid, block, 0, 0, a, c, b, n, 1); });
    return 0;
}
