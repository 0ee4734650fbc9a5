// fs_mgpcg.hip -- multigrid-preconditioned conjugate gradient pressure solve (optional, single GPU).
//
// Reference: CuSolver/Source/Solver.cu Conjugate :1226-1297, MG_Preconditioner :1312-1366,
// Smooth :848-860, SubRestrict/Restrict :822-834, 862-877, Prolongate :836-846, ComputeAp
// :789-799, UpdateResidual :801-810, UpdateP :812-820, InitConjugate :777-786, the shipped branch
// of Project (:1190-1192).  The operator is the 7-point Laplacian with ZERO outside the box
// (cg_sample), unlike the clamped Jacobi branch.
//
// The work arrays are dense like the reference's ([level][z][y][x], x fastest, no row padding);
// only div (input) and the pressure (output) live in the padded field layout.
//
// Deviations, all of them where the reference is not well defined:
//   * dot products: the reference's Mul + Reduce tree drops elements at most sizes and races
//     (Solver.cu:241-271, 1212-1224).  Here a dot product is the sum of the binary32 products,
//     accumulated in binary64 in a fixed order and narrowed to the float the host logic uses.
//   * z and the coarse levels of r are cleared completely before a V-cycle (the reference clears
//     mg_space/1024 whole blocks).
#include "fs_kernels.h"

#include <cstdlib>

#pragma clang fp contract(off)

namespace {

__host__ __device__ inline int cdiv(int a, int b) { return (a + b - 1) / b; }

struct Dim3i {
    int x, y, z;
};

// zero-outside sample of a dense array (cg_sample, Solver.cu:37-42)
__device__ __forceinline__ float cg(const float* __restrict__ f, int i, int j, int k, Dim3i d)
{
    if ((unsigned)i >= (unsigned)d.x || (unsigned)j >= (unsigned)d.y || (unsigned)k >= (unsigned)d.z) return 0.f;
    return f[(size_t)i + (size_t)j * d.x + (size_t)k * d.x * d.y];
}

// neibor_sum, Solver.cu:44-50, that association
__device__ __forceinline__ float neibor_sum(const float* __restrict__ f, int i, int j, int k, Dim3i d)
{
    return cg(f, i - 1, j, k, d) + cg(f, i + 1, j, k, d) + cg(f, i, j - 1, k, d) + cg(f, i, j + 1, k, d) +
           cg(f, i, j, k - 1, d) + cg(f, i, j, k + 1, d);
}

// InitConjugate: r = -div, x = 0
__global__ void k_cg_init(float* __restrict__ r, float* __restrict__ x, const float* __restrict__ div, Dim3i d,
                          int pitch, int plane)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y, k = blockIdx.z;
    if (i >= d.x || j >= d.y) return;
    const size_t ind = (size_t)i + (size_t)j * d.x + (size_t)k * d.x * d.y;
    r[ind] = -div[i + j * pitch + k * plane];
    x[ind] = 0.f;
}

__global__ void k_to_pressure(float* __restrict__ p, const float* __restrict__ x, Dim3i d, int pitch, int plane)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y, k = blockIdx.z;
    if (i >= d.x || j >= d.y) return;
    p[i + j * pitch + k * plane] = x[(size_t)i + (size_t)j * d.x + (size_t)k * d.x * d.y];
}

__global__ void k_apply_a(float* __restrict__ Ap, const float* __restrict__ p, Dim3i d)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y, k = blockIdx.z;
    if (i >= d.x || j >= d.y) return;
    const float pc = cg(p, i, j, k, d);
    Ap[(size_t)i + (size_t)j * d.x + (size_t)k * d.x * d.y] = 6.f * pc - neibor_sum(p, i, j, k, d);
}

__global__ void k_update_xr(float* __restrict__ x, float* __restrict__ r, const float* __restrict__ p,
                            const float* __restrict__ Ap, float alpha, size_t n)
{
    const size_t ind = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ind >= n) return;
    x[ind] += alpha * p[ind];
    r[ind] -= alpha * Ap[ind];
}

__global__ void k_update_p(float* __restrict__ p, const float* __restrict__ z, float beta, size_t n)
{
    const size_t ind = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ind >= n) return;
    p[ind] = z[ind] + beta * p[ind];
}

// One colour of red/black Gauss-Seidel: a thread per cell of that colour (the cells of the other
// colour are only read, so the sweep is race-free and order-independent).
__global__ void k_smooth(const float* __restrict__ r, float* z, int phase, Dim3i d)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y, k = blockIdx.z;
    const int i = 2 * t + ((j + k + phase) & 1);
    if (i >= d.x || j >= d.y) return;
    const size_t ind = (size_t)i + (size_t)j * d.x + (size_t)k * d.x * d.y;
    z[ind] = fsk::div6(r[ind] + neibor_sum(z, i, j, k, d));
}

// The eight SubRestrict launches of the reference, in their order, for one coarse cell: each
// adds half the residual of one child (double product and sum, narrowed on the store).
__global__ void k_restrict(const float* __restrict__ r, const float* __restrict__ z, float* __restrict__ rc, Dim3i d)
{
    const int ci = blockIdx.x * blockDim.x + threadIdx.x, cj = blockIdx.y * blockDim.y + threadIdx.y, ck = blockIdx.z;
    const int cx = d.x / 2, cy = d.y / 2;
    if (ci >= cx || cj >= cy) return;
    const size_t cind = (size_t)ci + (size_t)cj * cx + (size_t)ck * cx * cy;
    float acc = rc[cind];
#pragma unroll
    for (int phase = 0; phase < 8; ++phase) {
        const int i = ci * 2 + (phase & 1), j = cj * 2 + ((phase >> 1) & 1), k = ck * 2 + ((phase >> 2) & 1);
        const size_t ind = (size_t)i + (size_t)j * d.x + (size_t)k * d.x * d.y;
        const float res = r[ind] - (6 * z[ind] - neibor_sum(z, i, j, k, d));
        acc = (float)((double)acc + (double)res * 0.5);
    }
    rc[cind] = acc;
}

__global__ void k_prolongate(float* __restrict__ z, const float* __restrict__ zc, Dim3i d)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y, k = blockIdx.z;
    if (i >= d.x || j >= d.y) return;
    const size_t cind = (size_t)(i >> 1) + (size_t)(j >> 1) * (d.x >> 1) + (size_t)(k >> 1) * (d.x >> 1) * (d.y >> 1);
    z[(size_t)i + (size_t)j * d.x + (size_t)k * d.x * d.y] += zc[cind];
}

constexpr int kDotBlock = 256;

__device__ __forceinline__ double block_sum(double v, double* sh)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0)
        for (int q = 0; q < kDotBlock / 64; ++q) s += sh[q];
    return s;
}

// binary32 products, binary64 sums; every thread takes a fixed strided subset, every block writes
// one partial, the partials are summed by one block: the order never changes between runs
__global__ void __launch_bounds__(kDotBlock)
k_dot_partial(const float* __restrict__ a, const float* __restrict__ b, double* __restrict__ partial, size_t n)
{
    __shared__ double sh[kDotBlock / 64];
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * kDotBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kDotBlock) {
        const float prod = a[i] * b[i];
        s += (double)prod;
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ void __launch_bounds__(kDotBlock)
k_dot_final(const double* __restrict__ partial, int n, double* __restrict__ out)
{
    __shared__ double sh[kDotBlock / 64];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += kDotBlock) s += partial[i];
    s = block_sum(s, sh);
    if (threadIdx.x == 0) *out = s;
}

dim3 grid3(Dim3i d, int tx = 64, int ty = 4) { return dim3(cdiv(d.x, tx), cdiv(d.y, ty), d.z); }

}  // namespace

namespace fsk {

size_t mg_space(const FsGrid& g, int levels)
{
    size_t space = 0, t = (size_t)g.nx * g.ny * g.nz;
    for (int i = 0; i < levels; ++i) {
        space += t;
        t /= 8;
    }
    return space;
}

int cg_dot_partials() { return 1024; }

void cg_init(hipStream_t st, const FsGrid& g, float* r, float* x, const float* div)
{
    const Dim3i d{g.nx, g.ny, g.nz};
    hipLaunchKernelGGL(k_cg_init, grid3(d), dim3(64, 4), 0, st, r, x, div, d, g.pitch, g.plane);
}

void cg_to_pressure(hipStream_t st, const FsGrid& g, float* p, const float* x)
{
    const Dim3i d{g.nx, g.ny, g.nz};
    hipLaunchKernelGGL(k_to_pressure, grid3(d), dim3(64, 4), 0, st, p, x, d, g.pitch, g.plane);
}

void cg_apply_a(hipStream_t st, const FsGrid& g, float* Ap, const float* p)
{
    const Dim3i d{g.nx, g.ny, g.nz};
    hipLaunchKernelGGL(k_apply_a, grid3(d), dim3(64, 4), 0, st, Ap, p, d);
}

void cg_update_xr(hipStream_t st, const FsGrid& g, float* x, float* r, const float* p, const float* Ap, float alpha)
{
    const size_t n = (size_t)g.nx * g.ny * g.nz;
    hipLaunchKernelGGL(k_update_xr, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, r, p, Ap, alpha, n);
}

void cg_update_p(hipStream_t st, const FsGrid& g, float* p, const float* z, float beta)
{
    const size_t n = (size_t)g.nx * g.ny * g.nz;
    hipLaunchKernelGGL(k_update_p, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, z, beta, n);
}

void cg_dot(hipStream_t st, const FsGrid& g, const float* a, const float* b, double* partial, double* out)
{
    const size_t n = (size_t)g.nx * g.ny * g.nz;
    const int blocks = (int)((n + kDotBlock - 1) / kDotBlock < (size_t)cg_dot_partials() ? (n + kDotBlock - 1) / kDotBlock
                                                                                     : (size_t)cg_dot_partials());
    hipLaunchKernelGGL(k_dot_partial, dim3(blocks), dim3(kDotBlock), 0, st, a, b, partial, n);
    hipLaunchKernelGGL(k_dot_final, dim3(1), dim3(kDotBlock), 0, st, partial, blocks, out);
}

// Smallest plane (cells) whose smoothing goes through the temporally blocked kernel; below it a
// pass of that kernel has too few workgroups to beat four plain launches.  FS_MG_TB_MIN overrides
// (0 = every level the kernel can take, a huge value = never).
static long mg_tb_min_plane()
{
    if (const char* e = getenv("FS_MG_TB_MIN")) return atol(e);   // read per call: tests switch it
    return 128L * 128L;
}

// One V-cycle (Solver.cu:1312-1366): M^-1 r, level 0 of r is the input.  The smoother ping-pongs
// between z and zalt where it runs blocked, so the result is in either: the return value says
// which.  zalt needs no clearing (a blocked pass writes every cell of its output).
float* mg_precondition(hipStream_t st, const FsGrid& g, float* r, float* z, float* zalt, int levels,
                       int init_smooth_steps, int bottom_smooth_steps)
{
    const size_t space = mg_space(g, levels), n0 = (size_t)g.nx * g.ny * g.nz;
    (void)hipMemsetAsync(z, 0, space * sizeof(float), st);
    if (space > n0) (void)hipMemsetAsync(r + n0, 0, (space - n0) * sizeof(float), st);
    const long tb_min = mg_tb_min_plane();
    constexpr int kMaxLevels = 16;
    float* cur[kMaxLevels];      // array holding the current z of a level
    float* alt[kMaxLevels];
    Dim3i d{g.nx, g.ny, g.nz};
    size_t offset = 0;
    int l = 0;
    auto smooth = [&](int steps) {
        int phases = 2 * steps;
        if ((long)d.x * d.y >= tb_min) {
            while (phases >= 2) {
                const int take = phases >= 4 ? 4 : 2;
                if (!rbgs_pass(st, d.x, d.y, d.z, r + offset, cur[l], alt[l], take)) break;
                float* t = cur[l]; cur[l] = alt[l]; alt[l] = t;
                phases -= take;
            }
        }
        for (int i = 0; i < phases; ++i)   // red, black, red, ... (phases is even here)
            hipLaunchKernelGGL(k_smooth, dim3(cdiv(cdiv(d.x, 2), 32), cdiv(d.y, 4), d.z), dim3(32, 4), 0, st,
                               r + offset, cur[l], i & 1, d);
    };
    for (; l < levels - 1; ++l) {
        cur[l] = z + offset; alt[l] = zalt + offset;
        smooth(init_smooth_steps << l);
        const size_t n = (size_t)d.x * d.y * d.z;
        const Dim3i c{d.x / 2, d.y / 2, d.z / 2};
        hipLaunchKernelGGL(k_restrict, grid3(c, 32, 4), dim3(32, 4), 0, st, r + offset, cur[l], r + offset + n, d);
        offset += n;
        d = Dim3i{d.x >> 1, d.y >> 1, d.z >> 1};
    }
    cur[l] = z + offset; alt[l] = zalt + offset;
    smooth(bottom_smooth_steps);
    for (l = levels - 2; l >= 0; --l) {
        d = Dim3i{d.x << 1, d.y << 1, d.z << 1};
        const size_t n = (size_t)d.x * d.y * d.z;
        offset -= n;
        hipLaunchKernelGGL(k_prolongate, grid3(d), dim3(64, 4), 0, st, cur[l], cur[l + 1], d);
        smooth(init_smooth_steps << l);
    }
    return cur[0];
}

}  // namespace fsk
