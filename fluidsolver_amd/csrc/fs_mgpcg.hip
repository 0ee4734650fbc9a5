// fs_mgpcg.hip -- kernels and launchers of the multigrid-preconditioned conjugate gradient pressure
// solve (optional): the single-domain V-cycle and the per-window operations of a z-slab.
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
//   * every cell of z and of the coarse levels of r starts a V-cycle from zero: the restriction
//     starts its sums at zero and the first colour of a level writes both colours, so there are no
//     clearing passes (the reference clears mg_space/1024 whole blocks, not all of them).
#include "fs_kernels.h"

#include <cstdint>
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

using fsk::CgScalar;

__device__ __forceinline__ float cg_scalar(const CgScalar& c)
{
    return c.num ? (float)*c.num / (float)*c.den : c.v;
}

__global__ void k_update_xr(float* __restrict__ x, float* __restrict__ r, const float* __restrict__ p,
                            const float* __restrict__ Ap, CgScalar coef, size_t n)
{
    const size_t ind = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ind >= n) return;
    const float alpha = cg_scalar(coef);
    x[ind] += alpha * p[ind];
    r[ind] -= alpha * Ap[ind];
}

__global__ void k_update_p(float* __restrict__ p, const float* __restrict__ z, CgScalar coef, size_t n)
{
    const size_t ind = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ind >= n) return;
    const float beta = cg_scalar(coef);
    p[ind] = z[ind] + beta * p[ind];
}

// One colour of red/black Gauss-Seidel: a thread per cell of that colour (the cells of the other
// colour are only read, so the sweep is race-free and order-independent).
// (kpar: parity of the global z index of the array's plane 0 -- non-zero only for a slab's window)
__global__ void k_smooth(const float* __restrict__ r, float* z, int phase, Dim3i d, int kpar)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y, k = blockIdx.z;
    const int i = 2 * t + ((j + k + kpar + phase) & 1);
    if (i >= d.x || j >= d.y) return;
    const size_t ind = (size_t)i + (size_t)j * d.x + (size_t)k * d.x * d.y;
    z[ind] = fsk::div6(r[ind] + neibor_sum(z, i, j, k, d));
}

// Colour 0 of a level whose z is all zero, written together with the zeros of colour 1: a thread
// per pair of x-neighbouring cells.  The neighbour sum of zeros is kept in the expression (it turns
// a right-hand side of -0 into +0, as the general kernel does).
__global__ void k_smooth_first(const float* __restrict__ r, float* __restrict__ z, Dim3i d)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y, k = blockIdx.z;
    if (2 * t >= d.x || j >= d.y) return;
    const int q = (j + k) & 1;                    // the cell 2t + q has colour 0
    const size_t base = (size_t)2 * t + (size_t)j * d.x + (size_t)k * d.x * d.y;
    const float zero = 0.f;
    const float nsum = zero + zero + zero + zero + zero + zero;
    if (2 * t + q < d.x) z[base + q] = fsk::div6(r[base + q] + nsum);
    if (2 * t + (q ^ 1) < d.x) z[base + (q ^ 1)] = 0.f;
}

// All colours of `steps` sweeps of a level that fits one workgroup's LDS (z and the right-hand side,
// at most kSmoothLdsCells cells): the coarse levels of a V-cycle are a chain of tiny dependent
// launches otherwise.  Same expression and colour order as k_smooth.
constexpr int kSmoothLdsCells = 8192;
constexpr int kSmoothLdsThreads = 1024;

__global__ void __launch_bounds__(kSmoothLdsThreads)
k_smooth_lds(const float* __restrict__ r, float* __restrict__ z, int steps, Dim3i d)
{
    __shared__ float sz[kSmoothLdsCells];
    __shared__ float sr[kSmoothLdsCells];
    const int n = d.x * d.y * d.z, hx = (d.x + 1) >> 1, sy = d.x, sp = d.x * d.y;
    for (int c = threadIdx.x; c < n; c += kSmoothLdsThreads) {
        sz[c] = z[c];
        sr[c] = r[c];
    }
    __syncthreads();
    for (int ph = 0; ph < 2 * steps; ++ph) {
        for (int c = threadIdx.x; c < hx * d.y * d.z; c += kSmoothLdsThreads) {
            const int t = c % hx, rest = c / hx, j = rest % d.y, k = rest / d.y;
            const int i = 2 * t + ((j + k + ph) & 1);
            if (i >= d.x) continue;
            const int ind = i + j * sy + k * sp;
            const float xm = i > 0 ? sz[ind - 1] : 0.f, xp = i + 1 < d.x ? sz[ind + 1] : 0.f;
            const float ym = j > 0 ? sz[ind - sy] : 0.f, yp = j + 1 < d.y ? sz[ind + sy] : 0.f;
            const float km = k > 0 ? sz[ind - sp] : 0.f, kp = k + 1 < d.z ? sz[ind + sp] : 0.f;
            sz[ind] = fsk::div6(sr[ind] + (xm + xp + ym + yp + km + kp));
        }
        __syncthreads();
    }
    for (int c = threadIdx.x; c < n; c += kSmoothLdsThreads) z[c] = sz[c];
}

// The same for levels up to 16384 cells, with the work of a colour cut down to what it has to be.  z lives
// in LDS inside a shell of zeros (the zero-outside operator needs no bounds tests then); a thread
// owns up to ITEMS pairs of x-neighbouring cells for the whole launch - index, right-hand side and
// parity are worked out once and stay in registers - and every colour updates the one cell of each
// pair that has that colour: six LDS reads, the sum, the division, one LDS write.  (k_smooth_lds
// decomposes a running index with two divisions and two modulos per cell and colour, which is most
// of its time.)  Same expression and colour order as k_smooth.
constexpr int kSmoothPadFloats = 34 * 34 * 18 + 2048;   // 91 KB: a 32 x 32 x 16 level in its shell, or a 16 x 32 x 32 one

template <int ITEMS>
__global__ void __launch_bounds__(kSmoothLdsThreads)
k_smooth_pad(const float* __restrict__ r, float* __restrict__ z, int phases, int from_zero, Dim3i d)
{
    __shared__ float sz[kSmoothPadFloats];
    const int px = d.x + 2, pxy = px * (d.y + 2), padded = pxy * (d.z + 2);
    const int hx = (d.x + 1) >> 1, pairs = hx * d.y * d.z;
    for (int c = threadIdx.x; c < padded; c += kSmoothLdsThreads) sz[c] = 0.f;
    __syncthreads();
    int pb[ITEMS];          // padded index of the pair's first cell; -1: no such pair
    unsigned gi[ITEMS];     // its index in the level
    float r0[ITEMS], r1[ITEMS];
    int par[ITEMS];         // bit 0: parity of j + k, bit 1: the pair has a second cell
#pragma unroll
    for (int m = 0; m < ITEMS; ++m) {
        const int c = threadIdx.x + m * kSmoothLdsThreads;
        pb[m] = -1;
        gi[m] = 0;
        r0[m] = r1[m] = 0.f;
        par[m] = 0;
        if (c < pairs) {
            const int t = c % hx, rest = c / hx, j = rest % d.y, k = rest / d.y;
            const bool two = 2 * t + 1 < d.x;
            pb[m] = (2 * t + 1) + (j + 1) * px + (k + 1) * pxy;
            gi[m] = (unsigned)(2 * t + j * d.x + k * d.x * d.y);
            par[m] = ((j + k) & 1) | (two ? 2 : 0);
            r0[m] = r[gi[m]];
            if (two) r1[m] = r[gi[m] + 1];
            if (!from_zero) {
                sz[pb[m]] = z[gi[m]];
                if (two) sz[pb[m] + 1] = z[gi[m] + 1];
            }
        }
    }
    __syncthreads();
    for (int ph = 0; ph < phases; ++ph) {
#pragma unroll
        for (int m = 0; m < ITEMS; ++m) {
            const int q = (par[m] + ph) & 1;                  // which cell of the pair has this colour
            if (pb[m] < 0 || (q && !(par[m] & 2))) continue;
            const int ind = pb[m] + q;
            const float rhs = q ? r1[m] : r0[m];
            // x-1, x+1, y-1, y+1, z-1, z+1, summed left to right (neibor_sum)
            sz[ind] = fsk::div6(rhs + (sz[ind - 1] + sz[ind + 1] + sz[ind - px] + sz[ind + px] + sz[ind - pxy] + sz[ind + pxy]));
        }
        __syncthreads();
    }
#pragma unroll
    for (int m = 0; m < ITEMS; ++m) {
        if (pb[m] < 0) continue;
        z[gi[m]] = sz[pb[m]];
        if (par[m] & 2) z[gi[m] + 1] = sz[pb[m] + 1];
    }
}

// k_smooth_pad for a level of d: false if it does not fit (nothing launched)
static bool smooth_pad(hipStream_t st, const float* r, float* z, int phases, bool from_zero, Dim3i d)
{
    const long padded = (long)(d.x + 2) * (d.y + 2) * (d.z + 2);
    const long pairs = (long)((d.x + 1) / 2) * d.y * d.z;
    // (16 pairs per thread, a 32^3 level, would fit, but a colour then takes 4 us in the one
    //  workgroup against 3.4 us as a launch of its own that fills 64 CUs: 16384 cells is the limit)
    if (padded > kSmoothPadFloats || pairs > 8L * kSmoothLdsThreads) return false;
    const int items = (int)((pairs + kSmoothLdsThreads - 1) / kSmoothLdsThreads);
    const dim3 g(1), b(kSmoothLdsThreads);
    const int fz = from_zero ? 1 : 0;
    if (items <= 1) hipLaunchKernelGGL(k_smooth_pad<1>, g, b, 0, st, r, z, phases, fz, d);
    else if (items <= 2) hipLaunchKernelGGL(k_smooth_pad<2>, g, b, 0, st, r, z, phases, fz, d);
    else if (items <= 4) hipLaunchKernelGGL(k_smooth_pad<4>, g, b, 0, st, r, z, phases, fz, d);
    else hipLaunchKernelGGL(k_smooth_pad<8>, g, b, 0, st, r, z, phases, fz, d);
    return true;
}

// The eight SubRestrict launches of the reference, in their order, for one coarse cell: each
// adds half the residual of one child (double product and sum, narrowed on the store).
__global__ void k_restrict(const float* __restrict__ r, const float* __restrict__ z, float* __restrict__ rc, Dim3i d)
{
    const int ci = blockIdx.x * blockDim.x + threadIdx.x, cj = blockIdx.y * blockDim.y + threadIdx.y, ck = blockIdx.z;
    const int cx = d.x / 2, cy = d.y / 2;
    if (ci >= cx || cj >= cy) return;
    const size_t cind = (size_t)ci + (size_t)cj * cx + (size_t)ck * cx * cy;
    float acc = 0.f;   // (the reference accumulates onto a zeroed array)
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

// ---- z-slab windows -------------------------------------------------------------------------------
// On a slab a rank holds, of every level, the planes it owns plus ghost planes of its neighbours,
// as one dense array that the kernels above treat as a whole grid: beyond the array they read zero,
// which is the operator's boundary value at a global wall and a wrong value at a slab face.  That
// error enters the outermost ghost plane and moves one plane inward per stencil application; the
// driver (fs_project_mgpcg.hip) refreshes the ghost planes before it can reach an owned plane.

// Restriction (the eight SubRestrict launches, as k_restrict) into the OWNED planes of the coarse
// window: fine plane fk0 + 2*ck holds the lower children of owned coarse plane ck.
__global__ void k_restrict_win(const float* __restrict__ r, const float* __restrict__ z, float* __restrict__ rc,
                               Dim3i d, int fk0, int cz)
{
    const int ci = blockIdx.x * blockDim.x + threadIdx.x, cj = blockIdx.y * blockDim.y + threadIdx.y, ck = blockIdx.z;
    const int cx = d.x / 2, cy = d.y / 2;
    if (ci >= cx || cj >= cy || ck >= cz) return;
    const size_t cind = (size_t)ci + (size_t)cj * cx + (size_t)ck * cx * cy;
    float acc = 0.f;   // (the reference accumulates onto a zeroed array)
#pragma unroll
    for (int phase = 0; phase < 8; ++phase) {
        const int i = ci * 2 + (phase & 1), j = cj * 2 + ((phase >> 1) & 1), k = fk0 + ck * 2 + ((phase >> 2) & 1);
        const size_t ind = (size_t)i + (size_t)j * d.x + (size_t)k * d.x * d.y;
        const float res = r[ind] - (6 * z[ind] - neibor_sum(z, i, j, k, d));
        acc = (float)((double)acc + (double)res * 0.5);
    }
    rc[cind] = acc;
}

// Prolongation over every plane of the fine window whose parent plane the coarse window holds:
// fine plane k is global plane f0 + k, its parent global plane (f0 + k) >> 1 = coarse plane
// ((f0 + k) >> 1) - c0 of the coarse window (cheld planes).
__global__ void k_prolongate_win(float* __restrict__ z, const float* __restrict__ zc, Dim3i d, int f0, int c0,
                                 int cheld)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y, k = blockIdx.z;
    if (i >= d.x || j >= d.y) return;
    const int ck = ((f0 + k) >> 1) - c0;
    if (ck < 0 || ck >= cheld) return;
    const size_t cind = (size_t)(i >> 1) + (size_t)(j >> 1) * (d.x >> 1) + (size_t)ck * (d.x >> 1) * (d.y >> 1);
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

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

// k_apply_a and the partial sums of p . Ap in one pass over p; four cells along x per thread
// (nx % 4 == 0), a fixed strided set of quads per thread.
__global__ void __launch_bounds__(kDotBlock)
k_apply_a_dot4(float* __restrict__ Ap, const float* __restrict__ p, Dim3i d, double* __restrict__ partial, int k_lo,
               int k_hi)
{
    __shared__ double sh[kDotBlock / 64];
    const unsigned nx4 = (unsigned)d.x >> 2, items = nx4 * (unsigned)d.y * (unsigned)d.z;
    const size_t sy = (size_t)d.x, sz = (size_t)d.x * d.y;
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    double s = 0.0;
    for (unsigned it = blockIdx.x * kDotBlock + threadIdx.x; it < items; it += gridDim.x * kDotBlock) {
        const unsigned i4 = it % nx4, rest = it / nx4;
        const int j = (int)(rest % (unsigned)d.y), k = (int)(rest / (unsigned)d.y);
        const size_t ind = (size_t)4 * i4 + (size_t)j * sy + (size_t)k * sz;
        const f4 c = *(const f4*)(p + ind);
        const float lft = i4 > 0 ? p[ind - 1] : 0.f, rgt = i4 + 1 < nx4 ? p[ind + 4] : 0.f;
        const f4 dn = j > 0 ? *(const f4*)(p + ind - sy) : zero, up = j + 1 < d.y ? *(const f4*)(p + ind + sy) : zero;
        const f4 bk = k > 0 ? *(const f4*)(p + ind - sz) : zero, fw = k + 1 < d.z ? *(const f4*)(p + ind + sz) : zero;
        f4 a;
        a.x = 6.f * c.x - (lft + c.y + dn.x + up.x + bk.x + fw.x);
        a.y = 6.f * c.y - (c.x + c.z + dn.y + up.y + bk.y + fw.y);
        a.z = 6.f * c.z - (c.y + c.w + dn.z + up.z + bk.z + fw.z);
        a.w = 6.f * c.w - (c.z + rgt + dn.w + up.w + bk.w + fw.w);
        *(f4*)(Ap + ind) = a;
        if (k < k_lo || k >= k_hi) continue;   // (a slab's window: only its own planes count)
        s += (double)(c.x * a.x);
        s += (double)(c.y * a.y);
        s += (double)(c.z * a.z);
        s += (double)(c.w * a.w);
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// p' = z + beta p (Solver.cu:1290), A p' and the partial sums of p' . A p' in one pass: the new p of
// the six neighbours is formed again from z and the old p (the same expression, so the same values
// as the neighbours store), which is why p' goes to a second array.
__global__ void __launch_bounds__(kDotBlock)
k_update_p_apply_a_dot4(float* __restrict__ Ap, float* __restrict__ pn, const float* __restrict__ po,
                        const float* __restrict__ z, CgScalar coef, Dim3i d, double* __restrict__ partial)
{
    __shared__ double sh[kDotBlock / 64];
    const unsigned nx4 = (unsigned)d.x >> 2, items = nx4 * (unsigned)d.y * (unsigned)d.z;
    const size_t sy = (size_t)d.x, sz = (size_t)d.x * d.y;
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    const float beta = cg_scalar(coef);
    auto p1 = [&](size_t o) { return z[o] + beta * po[o]; };
    auto p4 = [&](size_t o) {
        const f4 zz = *(const f4*)(z + o), pp = *(const f4*)(po + o);
        return f4{zz.x + beta * pp.x, zz.y + beta * pp.y, zz.z + beta * pp.z, zz.w + beta * pp.w};
    };
    double s = 0.0;
    for (unsigned it = blockIdx.x * kDotBlock + threadIdx.x; it < items; it += gridDim.x * kDotBlock) {
        const unsigned i4 = it % nx4, rest = it / nx4;
        const int j = (int)(rest % (unsigned)d.y), k = (int)(rest / (unsigned)d.y);
        const size_t ind = (size_t)4 * i4 + (size_t)j * sy + (size_t)k * sz;
        const f4 c = p4(ind);
        const float lft = i4 > 0 ? p1(ind - 1) : 0.f, rgt = i4 + 1 < nx4 ? p1(ind + 4) : 0.f;
        const f4 dn = j > 0 ? p4(ind - sy) : zero, up = j + 1 < d.y ? p4(ind + sy) : zero;
        const f4 bk = k > 0 ? p4(ind - sz) : zero, fw = k + 1 < d.z ? p4(ind + sz) : zero;
        f4 a;
        a.x = 6.f * c.x - (lft + c.y + dn.x + up.x + bk.x + fw.x);
        a.y = 6.f * c.y - (c.x + c.z + dn.y + up.y + bk.y + fw.y);
        a.z = 6.f * c.z - (c.y + c.w + dn.z + up.z + bk.z + fw.z);
        a.w = 6.f * c.w - (c.z + rgt + dn.w + up.w + bk.w + fw.w);
        *(f4*)(pn + ind) = c;
        *(f4*)(Ap + ind) = a;
        s += (double)(c.x * a.x);
        s += (double)(c.y * a.y);
        s += (double)(c.z * a.z);
        s += (double)(c.w * a.w);
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// p' = z + beta p into another array
__global__ void k_update_p2(float* __restrict__ pn, const float* __restrict__ po, const float* __restrict__ z,
                            CgScalar coef, size_t n)
{
    const size_t ind = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ind >= n) return;
    const float beta = cg_scalar(coef);
    pn[ind] = z[ind] + beta * po[ind];
}

// k_update_xr and the partial sums of the new r . r
__global__ void __launch_bounds__(kDotBlock)
k_update_xr_dot4(float* __restrict__ x, float* __restrict__ r, const float* __restrict__ p,
                 const float* __restrict__ Ap, CgScalar coef, unsigned items, double* __restrict__ partial,
                 unsigned it_lo, unsigned it_hi)
{
    __shared__ double sh[kDotBlock / 64];
    double s = 0.0;
    const float alpha = cg_scalar(coef);
    for (unsigned it = blockIdx.x * kDotBlock + threadIdx.x; it < items; it += gridDim.x * kDotBlock) {
        const size_t ind = (size_t)4 * it;
        const f4 pv = *(const f4*)(p + ind), av = *(const f4*)(Ap + ind);
        f4 xv = *(const f4*)(x + ind), rv = *(const f4*)(r + ind);
        xv += alpha * pv;
        rv -= alpha * av;
        *(f4*)(x + ind) = xv;
        *(f4*)(r + ind) = rv;
        if (it < it_lo || it >= it_hi) continue;
        s += (double)(rv.x * rv.x);
        s += (double)(rv.y * rv.y);
        s += (double)(rv.z * rv.z);
        s += (double)(rv.w * rv.w);
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// k_dot_partial with 16-byte loads (n % 4 == 0)
__global__ void __launch_bounds__(kDotBlock)
k_dot_partial4(const float* __restrict__ a, const float* __restrict__ b, double* __restrict__ partial, unsigned items)
{
    __shared__ double sh[kDotBlock / 64];
    double s = 0.0;
    for (unsigned it = blockIdx.x * kDotBlock + threadIdx.x; it < items; it += gridDim.x * kDotBlock) {
        const f4 av = *(const f4*)(a + (size_t)4 * it), bv = *(const f4*)(b + (size_t)4 * it);
        s += (double)(av.x * bv.x);
        s += (double)(av.y * bv.y);
        s += (double)(av.z * bv.z);
        s += (double)(av.w * bv.w);
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// k_restrict with the rows of a coarse cell's 4 x 4 x 4 neighbourhood loaded once: 24 loads
// instead of 64 (nx % 2 == 0 keeps the 8-byte loads aligned)
__global__ void k_restrict2(const float* __restrict__ r, const float* __restrict__ z, float* __restrict__ rc, Dim3i d)
{
    const int ci = blockIdx.x * blockDim.x + threadIdx.x, cj = blockIdx.y * blockDim.y + threadIdx.y, ck = blockIdx.z;
    const int cx = d.x / 2, cy = d.y / 2;
    if (ci >= cx || cj >= cy) return;
    const size_t sy = (size_t)d.x, sz = (size_t)d.x * d.y;
    const int i = 2 * ci, j = 2 * cj, k = 2 * ck;
    const size_t base = (size_t)i + (size_t)j * sy + (size_t)k * sz;
    const f2 zero = {0.f, 0.f};
    // z of the children rows (y, z in {0,1}): cells i-1 .. i+2
    f2 zc[2][2];
    float zl[2][2], zr[2][2];
    f2 zyl[2], zyh[2], zzl[2], zzh[2], rr[2][2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const size_t o = base + a * sy + b * sz;
            zc[b][a] = *(const f2*)(z + o);
            rr[b][a] = *(const f2*)(r + o);
            zl[b][a] = i > 0 ? z[o - 1] : 0.f;
            zr[b][a] = i + 2 < d.x ? z[o + 2] : 0.f;
        }
        zyl[b] = j > 0 ? *(const f2*)(z + base + b * sz - sy) : zero;           // row y = j-1
        zyh[b] = j + 2 < d.y ? *(const f2*)(z + base + b * sz + 2 * sy) : zero;  // row y = j+2
        zzl[b] = k > 0 ? *(const f2*)(z + base + b * sy - sz) : zero;           // plane z = k-1, row y = j+b
        zzh[b] = k + 2 < d.z ? *(const f2*)(z + base + b * sy + 2 * sz) : zero;  // plane z = k+2
    }
    const size_t cind = (size_t)ci + (size_t)cj * cx + (size_t)ck * cx * cy;
    float acc = 0.f;   // (the reference accumulates onto a zeroed array)
#pragma unroll
    for (int phase = 0; phase < 8; ++phase) {
        const int a = phase & 1, b = (phase >> 1) & 1, c = (phase >> 2) & 1;   // child offsets in x, y, z
        const float zc0 = a ? zc[c][b].y : zc[c][b].x;
        const float xm = a ? zc[c][b].x : zl[c][b], xp = a ? zr[c][b] : zc[c][b].y;
        const f2 rowm = b ? zc[c][0] : zyl[c], rowp = b ? zyh[c] : zc[c][1];
        const f2 plm = c ? zc[0][b] : zzl[b], plp = c ? zzh[b] : zc[1][b];
        const float ym = a ? rowm.y : rowm.x, yp = a ? rowp.y : rowp.x;
        const float km = a ? plm.y : plm.x, kp = a ? plp.y : plp.x;
        const float rv = a ? rr[c][b].y : rr[c][b].x;
        const float res = rv - (6 * zc0 - (xm + xp + ym + yp + km + kp));
        acc = (float)((double)acc + (double)res * 0.5);
    }
    rc[cind] = acc;
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

void cg_update_xr(hipStream_t st, const FsGrid& g, float* x, float* r, const float* p, const float* Ap, CgScalar alpha)
{
    const size_t n = (size_t)g.nx * g.ny * g.nz;
    hipLaunchKernelGGL(k_update_xr, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, r, p, Ap, alpha, n);
}

static int dot_blocks(size_t work)
{
    const size_t b = (work + kDotBlock - 1) / kDotBlock;
    return (int)(b < (size_t)cg_dot_partials() ? b : (size_t)cg_dot_partials());
}

// Ap = A p and *out = p . Ap
void cg_apply_a_dot(hipStream_t st, const FsGrid& g, float* Ap, const float* p, double* partial, double* out, int k_lo,
                    int k_hi)
{
    if (k_hi < 0) k_hi = g.nz;
    if (g.nx % 4 != 0) {
        cg_apply_a(st, g, Ap, p);
        FsGrid go = g;
        go.nz = k_hi - k_lo;
        const size_t o = (size_t)k_lo * g.nx * g.ny;
        cg_dot(st, go, p + o, Ap + o, partial, out);
        return;
    }
    const Dim3i d{g.nx, g.ny, g.nz};
    const int blocks = dot_blocks((size_t)g.nx / 4 * g.ny * g.nz);
    hipLaunchKernelGGL(k_apply_a_dot4, dim3(blocks), dim3(kDotBlock), 0, st, Ap, p, d, partial, k_lo, k_hi);
    hipLaunchKernelGGL(k_dot_final, dim3(1), dim3(kDotBlock), 0, st, partial, blocks, out);
}

// pn = z + beta po, Ap = A pn and *out = pn . Ap
void cg_update_p_apply_a_dot(hipStream_t st, const FsGrid& g, float* Ap, float* pn, const float* po, const float* z,
                             CgScalar beta, double* partial, double* out)
{
    if (g.nx % 4 != 0) {
        const size_t n = (size_t)g.nx * g.ny * g.nz;
        hipLaunchKernelGGL(k_update_p2, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, pn, po, z, beta, n);
        cg_apply_a(st, g, Ap, pn);
        cg_dot(st, g, pn, Ap, partial, out);
        return;
    }
    const Dim3i d{g.nx, g.ny, g.nz};
    const int blocks = dot_blocks((size_t)g.nx / 4 * g.ny * g.nz);
    hipLaunchKernelGGL(k_update_p_apply_a_dot4, dim3(blocks), dim3(kDotBlock), 0, st, Ap, pn, po, z, beta, d, partial);
    hipLaunchKernelGGL(k_dot_final, dim3(1), dim3(kDotBlock), 0, st, partial, blocks, out);
}

// x += alpha p, r -= alpha Ap and *out = r . r of the new r
void cg_update_xr_dot(hipStream_t st, const FsGrid& g, float* x, float* r, const float* p, const float* Ap,
                      CgScalar alpha, double* partial, double* out, int k_lo, int k_hi)
{
    if (k_hi < 0) k_hi = g.nz;
    const size_t n = (size_t)g.nx * g.ny * g.nz, cell = (size_t)g.nx * g.ny;
    if (n % 4 != 0 || cell % 4 != 0) {
        cg_update_xr(st, g, x, r, p, Ap, alpha);
        FsGrid go = g;
        go.nz = k_hi - k_lo;
        cg_dot(st, go, r + k_lo * cell, r + k_lo * cell, partial, out);
        return;
    }
    const int blocks = dot_blocks(n / 4);
    hipLaunchKernelGGL(k_update_xr_dot4, dim3(blocks), dim3(kDotBlock), 0, st, x, r, p, Ap, alpha, (unsigned)(n / 4),
                       partial, (unsigned)(k_lo * cell / 4), (unsigned)(k_hi * cell / 4));
    hipLaunchKernelGGL(k_dot_final, dim3(1), dim3(kDotBlock), 0, st, partial, blocks, out);
}

void cg_update_p(hipStream_t st, const FsGrid& g, float* p, const float* z, CgScalar beta)
{
    const size_t n = (size_t)g.nx * g.ny * g.nz;
    hipLaunchKernelGGL(k_update_p, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, z, beta, n);
}

void cg_dot(hipStream_t st, const FsGrid& g, const float* a, const float* b, double* partial, double* out)
{
    const size_t n = (size_t)g.nx * g.ny * g.nz;
    const bool quads = n % 4 == 0 && (((uintptr_t)a | (uintptr_t)b) & 15u) == 0;
    const int blocks = dot_blocks(quads ? n / 4 : n);
    if (quads) hipLaunchKernelGGL(k_dot_partial4, dim3(blocks), dim3(kDotBlock), 0, st, a, b, partial, (unsigned)(n / 4));
    else hipLaunchKernelGGL(k_dot_partial, dim3(blocks), dim3(kDotBlock), 0, st, a, b, partial, n);
    hipLaunchKernelGGL(k_dot_final, dim3(1), dim3(kDotBlock), 0, st, partial, blocks, out);
}

// Smallest plane (cells) whose smoothing goes through the temporally blocked kernel; below it a
// pass of that kernel has too few workgroups to beat four plain launches.  FS_MG_TB_MIN overrides
// (0 = every level the kernel can take, a huge value = never).
// (FsEnv::mg_tb_min; FS_MG_LDS_PAD=0 selects the older single-workgroup smoother, FS_MG_NO_LDS=1
// forces the per-colour launches on small levels: tests)

// One V-cycle (Solver.cu:1312-1366): M^-1 r, level 0 of r is the input.  The smoother ping-pongs
// between z and zalt where it runs blocked, so the result is in either: the return value says
// which.  zalt needs no clearing (a blocked pass writes every cell of its output).
float* mg_precondition(hipStream_t st, const FsGrid& g, float* r, float* z, float* zalt, int levels,
                       int init_smooth_steps, int bottom_smooth_steps)
{
    // (the coarse levels of r need no clearing: the restriction writes every cell of them)
    const long tb_min = g.env.mg_tb_min;
    const bool no_lds = g.env.mg_no_lds != 0, pad = g.env.mg_lds_pad != 0;
    constexpr int kMaxLevels = 16;
    float* cur[kMaxLevels];      // array holding the current z of a level
    float* alt[kMaxLevels];
    Dim3i d{g.nx, g.ny, g.nz};
    size_t offset = 0;
    int l = 0;
    // z starts at zero on every level: a blocked first pass takes that as known instead of
    // reading it, any other smoother gets the level cleared first
    auto smooth = [&](int steps, bool from_zero) {
        int phases = 2 * steps;
        if ((long)d.x * d.y >= tb_min) {
            while (phases >= 2) {
                const int take = phases >= 4 ? 4 : 2;
                if (!rbgs_pass(st, g.env, d.x, d.y, d.z, r + offset, cur[l], alt[l], take, from_zero)) break;
                float* t = cur[l]; cur[l] = alt[l]; alt[l] = t;
                phases -= take;
                from_zero = false;
            }
        }
        // levels up to 16384 cells: every colour in one workgroup (FS_MG_LDS_PAD=0: up to 8192 cells,
        // through the older kernel)
        if (phases > 0 && !no_lds && pad &&
            smooth_pad(st, r + offset, cur[l], phases, from_zero, d))
            return;
        const dim3 cgrid(cdiv(cdiv(d.x, 2), 32), cdiv(d.y, 4), d.z), cblock(32, 4);
        if (phases > 0 && (size_t)d.x * d.y * d.z <= (size_t)kSmoothLdsCells && !no_lds) {
            if (from_zero) (void)hipMemsetAsync(cur[l], 0, (size_t)d.x * d.y * d.z * sizeof(float), st);
            hipLaunchKernelGGL(k_smooth_lds, dim3(1), dim3(kSmoothLdsThreads), 0, st, r + offset, cur[l], phases / 2, d);
            return;
        }
        int i = 0;
        if (from_zero && phases > 0) {
            // the first colour of a level that is all zero also writes the zeros of the other colour:
            // no clearing pass
            hipLaunchKernelGGL(k_smooth_first, cgrid, cblock, 0, st, r + offset, cur[l], d);
            i = 1;
        } else if (from_zero) {
            (void)hipMemsetAsync(cur[l], 0, (size_t)d.x * d.y * d.z * sizeof(float), st);
        }
        for (; i < phases; ++i)   // red, black, red, ... (phases is even here)
            hipLaunchKernelGGL(k_smooth, cgrid, cblock, 0, st, r + offset, cur[l], i & 1, d, 0);
    };
    for (; l < levels - 1; ++l) {
        cur[l] = z + offset; alt[l] = zalt + offset;
        smooth(init_smooth_steps << l, true);
        const size_t n = (size_t)d.x * d.y * d.z;
        const Dim3i c{d.x / 2, d.y / 2, d.z / 2};
        // (the level offsets are even, so 8-byte loads are aligned whenever nx is even)
        if (d.x % 2 == 0 && d.y % 2 == 0 && d.z % 2 == 0 && (n & 1) == 0 && (offset & 1) == 0)
            hipLaunchKernelGGL(k_restrict2, grid3(c, 32, 4), dim3(32, 4), 0, st, r + offset, cur[l], r + offset + n, d);
        else
            hipLaunchKernelGGL(k_restrict, grid3(c, 32, 4), dim3(32, 4), 0, st, r + offset, cur[l], r + offset + n, d);
        offset += n;
        d = Dim3i{d.x >> 1, d.y >> 1, d.z >> 1};
    }
    cur[l] = z + offset; alt[l] = zalt + offset;
    smooth(bottom_smooth_steps, true);
    for (l = levels - 2; l >= 0; --l) {
        d = Dim3i{d.x << 1, d.y << 1, d.z << 1};
        const size_t n = (size_t)d.x * d.y * d.z;
        offset -= n;
        hipLaunchKernelGGL(k_prolongate, grid3(d), dim3(64, 4), 0, st, cur[l], cur[l + 1], d);
        smooth(init_smooth_steps << l, false);
    }
    return cur[0];
}

// ---- launchers for the z-slab windows (dense arrays of nx x ny x held planes) ------------------------

void mgs_smooth_phase(hipStream_t st, const float* r, float* z, int nx, int ny, int held, int phase, int kpar)
{
    const Dim3i d{nx, ny, held};
    hipLaunchKernelGGL(k_smooth, dim3(cdiv(cdiv(nx, 2), 32), cdiv(ny, 4), held), dim3(32, 4), 0, st, r, z, phase & 1, d,
                       kpar & 1);
}

// `colours` (2 or 4, red first) colours of the window in one blocked pass from z into zalt; false if
// the level is not one the blocked kernel is used for (nothing launched)
bool mgs_smooth_pass(hipStream_t st, const FsEnv& env, const float* r, const float* z, float* zalt, int nx, int ny,
                     int held, int colours, bool z_is_zero, int kpar)
{
    if ((long)nx * ny < env.mg_tb_min) return false;
    return rbgs_pass(st, env, nx, ny, held, r, z, zalt, colours, z_is_zero, kpar);
}

void mgs_restrict(hipStream_t st, const float* rf, const float* zf, int nx, int ny, int held_f, int fk0,
                  float* rc_owned, int cz)
{
    const Dim3i d{nx, ny, held_f};
    hipLaunchKernelGGL(k_restrict_win, dim3(cdiv(nx / 2, 32), cdiv(ny / 2, 4), cz), dim3(32, 4), 0, st, rf, zf, rc_owned,
                       d, fk0, cz);
}

void mgs_prolongate(hipStream_t st, float* zf, int nx, int ny, int held_f, int f0, const float* zc, int c0, int cheld)
{
    const Dim3i d{nx, ny, held_f};
    hipLaunchKernelGGL(k_prolongate_win, grid3(d), dim3(64, 4), 0, st, zf, zc, d, f0, c0, cheld);
}

}  // namespace fsk
