// fs_vort.hip -- vorticity confinement (optional stage), defined-behaviour variant.
//
// Reference: CuSolver/Source/Solver.cu FindVortx/y/z :485-525 (launched :1035-1037, before the
// advection) and ApplyVortConfinement :540-605 (launched :1078-1080, after it).  The reference
// writes each vorticity array with cell strides (nx, nx*ny) into nx*ny*nz floats but gathers from
// it with the strides and clamps of an (nx,ny+1,nz+1) / (nx+1,ny,nz+1) / (nx+1,ny+1,nz) array
// (:556-558), which runs past the allocation.  Here each array holds (nx+1)(ny+1)(nz+1) floats,
// zeroed once at creation and written only in its first nx*ny*nz, so the same index arithmetic
// stays in bounds and finds either a vorticity value or zero.  Everything else is literal:
// mismatched strides, the launch-dimension-dependent guard (:548), float sqrt for the magnitudes,
// the double-precision sqrt(... + 1e-5) (:593), true divisions.
//
// The arrays are dense (no row padding): the reference's flat indices are used as they are.
// On a z-slab a rank holds the window of that flat index space its own faces can reach (its
// cells' entries plus halo_planes() xy-planes of nx*ny entries on each side, filled by the
// neighbours): a gather at face (i, j, k) lands up to (j + k) / ny entries-planes above plane k
// because of the mismatched strides.  The pointers the kernels receive are offset so that the
// global flat index addresses the window.
#include "fs_kernels.h"

#pragma clang fp contract(off)

namespace {

__host__ __device__ inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// clamped point sample of a staggered velocity component in the padded device layout
// (sample(), Solver.cu:30-34)
__device__ __forceinline__ float vel(const float* __restrict__ f, int x, int y, int z, int dx, int dy,
                                     int dz, int pitch, int plane)
{
    x = max(0, min(x, dx - 1));
    y = max(0, min(y, dy - 1));
    z = max(0, min(z, dz - 1));
    return f[x + y * pitch + z * plane];
}

// Solver.cu:485-525, all three components in one pass
__global__ void k_find_vort(float* __restrict__ vortx, float* __restrict__ vorty, float* __restrict__ vortz,
                            const float* __restrict__ ux, const float* __restrict__ uy,
                            const float* __restrict__ uz, int nx, int ny, int nz, int pitch, int plane, int zb)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y * blockDim.y + threadIdx.y;
    const int k = zb + blockIdx.z;
    if (i >= nx || j >= ny) return;
    const size_t ind = (size_t)i + (size_t)j * nx + (size_t)k * nx * ny;
    {   // :492-496
        float ubh = vel(uy, i, j, k - 1, nx, ny + 1, nz, pitch, plane);
        float uf = vel(uy, i, j, k, nx, ny + 1, nz, pitch, plane);
        float ubo = vel(uz, i, j - 1, k, nx, ny, nz + 1, pitch, plane);
        float ut = vel(uz, i, j, k, nx, ny, nz + 1, pitch, plane);
        vortx[ind] = uf - ubh - ut + ubo;
    }
    {   // :506-510
        float ul = vel(uz, i - 1, j, k, nx, ny, nz + 1, pitch, plane);
        float ur = vel(uz, i, j, k, nx, ny, nz + 1, pitch, plane);
        float ubh = vel(ux, i, j, k - 1, nx + 1, ny, nz, pitch, plane);
        float uf = vel(ux, i, j, k, nx + 1, ny, nz, pitch, plane);
        vorty[ind] = ur - ul - uf + ubh;
    }
    {   // :520-524
        float ul = vel(uy, i - 1, j, k, nx, ny + 1, nz, pitch, plane);
        float ur = vel(uy, i, j, k, nx, ny + 1, nz, pitch, plane);
        float ubo = vel(ux, i, j - 1, k, nx + 1, ny, nz, pitch, plane);
        float ut = vel(ux, i, j, k, nx + 1, ny, nz, pitch, plane);
        vortz[ind] = ut - ubo - ur + ul;
    }
}

// One vorticity array as the gathers see it: dims (dx,dy,dz) give both the clamp and the strides.
struct VortView {
    const float* __restrict__ f;
    int dx, dy, dz;
};

// Index contributions of the taps i-1, i, i+1 (and j, k alike) of one view, clamped like sample().
struct VortTaps {
    int x[3], y[3], z[3];
};

__device__ __forceinline__ VortTaps make_taps(const VortView& v, int i, int j, int k)
{
    VortTaps t;
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        t.x[o] = max(0, min(i + o - 1, v.dx - 1));
        t.y[o] = max(0, min(j + o - 1, v.dy - 1)) * v.dx;
        t.z[o] = max(0, min(k + o - 1, v.dz - 1)) * v.dx * v.dy;
    }
    return t;
}

// lerp(a, b, s) = a*(1-s) + b*s (Math.cuh:57-60) for the two weights that occur here.  With s = 0
// the result is a for every finite b (a*1 + b*0), so b is not loaded.  FM: the left product fused into
// the sum, what a contracting compiler makes of the expression (FS_FLAG_FMAD; the halves are exact
// for normal numbers, so the two forms differ for subnormal operands only).
template <bool HALF, bool FM>
__device__ __forceinline__ float lerp_s(float a, float b)
{
    if constexpr (!HALF) return a;
    else if constexpr (FM) return __builtin_fmaf(a, 1.0f - 0.5f, b * 0.5f);
    else return a * (1.0f - 0.5f) + b * 0.5f;
}

// trilerp (Solver.cu:52-113, clamped samples) at a point whose coordinates, after the array's
// staggering offset is subtracted, are (i,j,k) + (HX,HY,HZ)/2 with i,j,k >= 3: all coordinates are
// positive, so int() and floor() agree, the lower tap is i + floor(H/2) and the weight is 0 or
// one half according to the parity of H -- both known at compile time.  Identical taps of
// different gathers are loaded once (same address expression, no store in between).
template <int HX, int HY, int HZ, bool FM>
__device__ __forceinline__ float gather(const VortView& v, const VortTaps& t)
{
    constexpr int ox = (HX >> 1) + 1, oy = (HY >> 1) + 1, oz = (HZ >> 1) + 1;   // 0..2, arithmetic shift = floor
    constexpr bool fx = HX & 1, fy = HY & 1, fz = HZ & 1;
    static_assert(ox >= 0 && ox + fx <= 2 && oy >= 0 && oy + fy <= 2 && oz >= 0 && oz + fz <= 2, "tap range");
    auto tap = [&](int a, int b, int c) { return v.f[t.x[a] + t.y[b] + t.z[c]]; };
    auto row = [&](int b, int c) {
        if constexpr (fx) return lerp_s<true, FM>(tap(ox, b, c), tap(ox + 1, b, c));
        else return tap(ox, b, c);
    };
    auto slab = [&](int c) {
        if constexpr (fy) return lerp_s<true, FM>(row(oy, c), row(oy + 1, c));
        else return row(oy, c);
    };
    if constexpr (fz) return lerp_s<true, FM>(slab(oz), slab(oz + 1));
    else return slab(oz);
}

// |omega| at face position + (DX,DY,DZ)/2, Solver.cu:556-584.  PX,PY,PZ: twice the face's
// staggering (0 on the component's own axis, 1 elsewhere); the arrays are staggered by one half
// along x (vortx), y (vorty), z (vortz).
template <int PX, int PY, int PZ, int DX, int DY, int DZ, bool FM>
__device__ __forceinline__ void vort3(const VortView& vx, const VortTaps& tx, const VortView& vy,
                                      const VortTaps& ty, const VortView& vz, const VortTaps& tz,
                                      float& ox, float& oy, float& oz)
{
    ox = gather<PX + DX - 1, PY + DY, PZ + DZ, FM>(vx, tx);
    oy = gather<PX + DX, PY + DY - 1, PZ + DZ, FM>(vy, ty);
    oz = gather<PX + DX, PY + DY, PZ + DZ - 1, FM>(vz, tz);
}

template <int PX, int PY, int PZ, int DX, int DY, int DZ, bool FM>
__device__ __forceinline__ float vort_norm(const VortView& vx, const VortTaps& tx, const VortView& vy,
                                           const VortTaps& ty, const VortView& vz, const VortTaps& tz)
{
    float a, b, c;
    vort3<PX, PY, PZ, DX, DY, DZ, FM>(vx, tx, vy, ty, vz, tz, a, b, c);
    return sqrtf(a * a + b * b + c * c);
}

// Solver.cu:540-605 for component DIR (1: ux, 2: uy, 3: uz); (dx,dy,dz) are the dims of that
// component, which are also the reference's launch dims, so the guard of :548 is literal.
template <int DIR, bool FM>
__global__ void k_vort_confine(float* __restrict__ u, VortView vx, VortView vy, VortView vz, int dx, int dy,
                               int dz, int pitch, int plane, float coeff, int zb)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y * blockDim.y + threadIdx.y;
    const int k = zb + blockIdx.z;
    if (!(i > 2 && i < dx - 2 && j > 2 && j < dy - 2 && k > 2 && k < dz - 2)) return;
    constexpr int PX = DIR == 1 ? 0 : 1, PY = DIR == 2 ? 0 : 1, PZ = DIR == 3 ? 0 : 1;
    const VortTaps tx = make_taps(vx, i, j, k), ty = make_taps(vy, i, j, k), tz = make_taps(vz, i, j, k);

    const float vt = vort_norm<PX, PY, PZ, 0, 1, 0, FM>(vx, tx, vy, ty, vz, tz);
    const float vbo = vort_norm<PX, PY, PZ, 0, -1, 0, FM>(vx, tx, vy, ty, vz, tz);
    const float vr = vort_norm<PX, PY, PZ, 1, 0, 0, FM>(vx, tx, vy, ty, vz, tz);
    const float vl = vort_norm<PX, PY, PZ, -1, 0, 0, FM>(vx, tx, vy, ty, vz, tz);
    const float vf = vort_norm<PX, PY, PZ, 0, 0, 1, FM>(vx, tx, vy, ty, vz, tz);
    const float vbh = vort_norm<PX, PY, PZ, 0, 0, -1, FM>(vx, tx, vy, ty, vz, tz);
    float vxc, vyc, vzc;
    vort3<PX, PY, PZ, 0, 0, 0, FM>(vx, tx, vy, ty, vz, tz, vxc, vyc, vzc);

    float gx = vr - vl;
    float gy = vt - vbo;
    float gz = vf - vbh;
    // :593 -- the float sum meets a double literal: widened, double sqrt, narrowed on assignment
    const float n = (float)sqrt((double)(gx * gx + gy * gy + gz * gz) + 1e-5);
    gx /= n;
    gy /= n;
    gz /= n;
    float* dst = u + i + j * pitch + k * plane;
    if (DIR == 1) *dst += coeff * (gy * vzc - gz * vyc);
    else if (DIR == 2) *dst += coeff * (gz * vxc - gx * vzc);
    else *dst += coeff * (gx * vyc - gy * vxc);
}

// ---- the same confinement with every |omega| evaluated once --------------------------------------
// The six off-centre |omega| of a face lie half a cell from it along each axis.  Over the faces of
// all three components those points form four lattices, each indexed like the cells/faces (i, j, k):
//   A  (i+1/2, j+1/2, k+1/2)  cell centres          ux: i-1, i    uy: j-1, j    uz: k-1, k
//   B  (i,     j,     k+1/2)  z-edge midpoints      ux: j, j+1    uy: i, i+1
//   C  (i,     j+1/2, k    )  y-edge midpoints      ux: k, k+1    uz: i, i+1
//   D  (i+1/2, j,     k    )  x-edge midpoints      uy: k, k+1    uz: j, j+1
// (coordinates in cells; the taps and weights of a gather depend on the point alone, so a value read
// from a lattice is bit for bit the value the face would have gathered itself).  k_vort_lattice
// evaluates them for the planes the faces of this rank touch, k_vort_confine_all applies the force to
// the three components from them: 4 instead of 18 evaluations per cell.
struct Lattice {
    float *A, *B, *C, *D;   // (nx+1) x (ny+1) x planes each, plane 0 = global plane k0
    int sx, sxy, k0;
    __device__ __forceinline__ size_t at(int i, int j, int k) const
    {
        return (size_t)i + (size_t)j * sx + (size_t)(k - k0) * sxy;
    }
};

template <bool FM>
__global__ void k_vort_lattice(Lattice Lt, VortView vx, VortView vy, VortView vz, int nx, int ny, int kb)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y * blockDim.y + threadIdx.y;
    const int k = kb + blockIdx.z;
    if (i > nx || j > ny) return;
    const VortTaps tx = make_taps(vx, i, j, k), ty = make_taps(vy, i, j, k), tz = make_taps(vz, i, j, k);
    const size_t o = Lt.at(i, j, k);
    // (P + D of vort_norm is twice the point's offset from (i, j, k))
    Lt.A[o] = vort_norm<0, 1, 1, 1, 0, 0, FM>(vx, tx, vy, ty, vz, tz);
    Lt.B[o] = vort_norm<0, 1, 1, 0, -1, 0, FM>(vx, tx, vy, ty, vz, tz);
    Lt.C[o] = vort_norm<0, 1, 1, 0, 0, -1, FM>(vx, tx, vy, ty, vz, tz);
    Lt.D[o] = vort_norm<1, 0, 1, 0, 0, -1, FM>(vx, tx, vy, ty, vz, tz);
}

// Solver.cu:586-604 for one face from its six |omega| and the vorticity at the face itself
template <int DIR>
__device__ __forceinline__ void confine_face(float* dst, float coeff, float vr, float vl, float vt, float vbo, float vf,
                                             float vbh, float vxc, float vyc, float vzc)
{
    float gx = vr - vl;
    float gy = vt - vbo;
    float gz = vf - vbh;
    // :593 -- the float sum meets a double literal: widened, double sqrt, narrowed on assignment
    const float n = (float)sqrt((double)(gx * gx + gy * gy + gz * gz) + 1e-5);
    gx /= n;
    gy /= n;
    gz /= n;
    if (DIR == 1) *dst += coeff * (gy * vzc - gz * vyc);
    else if (DIR == 2) *dst += coeff * (gz * vxc - gx * vzc);
    else *dst += coeff * (gx * vyc - gy * vxc);
}

// All three components of the faces (i, j, k), k in [kb, kb + gridDim.z); the guards are those of the
// reference's three launches (:548 with each component's own dims); kxy_end: first plane without ux
// and uy faces of this rank (the shared uz face plane of a slab has none)
template <bool FM>
__global__ void k_vort_confine_all(float* __restrict__ ux, float* __restrict__ uy, float* __restrict__ uz, Lattice Lt,
                                   VortView vx, VortView vy, VortView vz, int nx, int ny, int nz, int pitch, int plane,
                                   float coeff, int kb, int kxy_end)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y * blockDim.y + threadIdx.y;
    const int k = kb + blockIdx.z;
    if (i > nx || j > ny || i < 3 || j < 3 || k < 3) return;
    const VortTaps tx = make_taps(vx, i, j, k), ty = make_taps(vy, i, j, k), tz = make_taps(vz, i, j, k);
    const size_t o = Lt.at(i, j, k);
    const int off = i + j * pitch + k * plane;
    float a, b, c;
    if (k < kxy_end && i < nx + 1 - 2 && j < ny - 2 && k < nz - 2) {          // ux: dims (nx+1, ny, nz)
        vort3<0, 1, 1, 0, 0, 0, FM>(vx, tx, vy, ty, vz, tz, a, b, c);
        confine_face<1>(ux + off, coeff, Lt.A[o], Lt.A[o - 1], Lt.B[o + Lt.sx], Lt.B[o], Lt.C[o + Lt.sxy], Lt.C[o], a, b, c);
    }
    if (k < kxy_end && i < nx - 2 && j < ny + 1 - 2 && k < nz - 2) {          // uy: dims (nx, ny+1, nz)
        vort3<1, 0, 1, 0, 0, 0, FM>(vx, tx, vy, ty, vz, tz, a, b, c);
        confine_face<2>(uy + off, coeff, Lt.B[o + 1], Lt.B[o], Lt.A[o], Lt.A[o - Lt.sx], Lt.D[o + Lt.sxy], Lt.D[o], a, b, c);
    }
    if (i < nx - 2 && j < ny - 2 && k < nz + 1 - 2) {                          // uz: dims (nx, ny, nz+1)
        vort3<1, 1, 0, 0, 0, 0, FM>(vx, tx, vy, ty, vz, tz, a, b, c);
        confine_face<3>(uz + off, coeff, Lt.C[o + 1], Lt.C[o], Lt.D[o + Lt.sx], Lt.D[o], Lt.A[o], Lt.A[o - Lt.sxy], a, b, c);
    }
}

}  // namespace

namespace fsk {

// xy-planes (nx*ny entries each) of the vorticity arrays a rank needs beyond its own cells: one
// below (the z - 1 taps), and above as far as the largest flat index the gathers of its topmost
// faces form, (z1 + 1) * (nx+1)(ny+1) + ..., exceeds z1 * nx*ny.  One figure for all ranks (the
// topmost one's), so every exchange has the same size.
int vort_halo_planes(const FsGrid& g)
{
    const long long cell = (long long)g.nx * g.ny;
    const long long top = ((long long)g.nz + 2) * ((long long)g.nx + 1) * ((long long)g.ny + 1);
    return (int)((top - (long long)g.nz * cell + cell - 1) / cell);
}

// entries one array of this rank holds: its cells, the halo planes on each non-wall side and,
// at the upper wall, what the single-domain layout holds beyond the cells
size_t vort_elems(const FsGrid& g)
{
    const size_t cell = (size_t)g.nx * g.ny;
    const int h = vort_halo_planes(g);
    return cell * ((size_t)g.nzl + (g.wall_lo ? 0 : h) + h);
}

// flat index of the first entry held
size_t vort_first(const FsGrid& g)
{
    return (size_t)g.nx * g.ny * (size_t)(g.z0 - (g.wall_lo ? 0 : vort_halo_planes(g)));
}

void find_vort(hipStream_t st, const FsGrid& g, float* vortx, float* vorty, float* vortz, const float* ux,
               const float* uy, const float* uz)
{
    // this rank's cells; the window pointers and the velocity take global indices
    const ptrdiff_t w = (ptrdiff_t)vort_first(g), zoff = (ptrdiff_t)g.z0 * g.plane;
    dim3 block(64, 4), grid(cdiv(g.nx, 64), cdiv(g.ny, 4), g.nzl);
    hipLaunchKernelGGL(k_find_vort, grid, block, 0, st, vortx - w, vorty - w, vortz - w, ux - zoff, uy - zoff,
                       uz - zoff, g.nx, g.ny, g.nz, g.pitch, g.plane, g.z0);
}

// Planes of the |omega| lattices a rank holds: z0 - 1 (the cell centres below its first uz faces)
// up to z0 + nzl (the planes above its last ux / uy faces, the plane of the shared uz faces).  The
// gathers of those lattice points reach the vorticity planes z0 - 2 .. z0 + nzl + 1: at the top what
// the faces' own gathers reach, at the bottom one plane more, which the window's halo also holds.
static void lattice_planes(const FsGrid& g, int* kb, int* ke)
{
    *kb = max(g.z0 - 1, 0);
    *ke = g.z0 + g.nzl + 1;
}

size_t vort_lattice_elems(const FsGrid& g)
{
    int kb, ke;
    lattice_planes(g, &kb, &ke);
    return 4 * (size_t)(g.nx + 1) * (g.ny + 1) * (size_t)(ke - kb);
}

void vort_confine(hipStream_t st, const FsGrid& g, float* ux, float* uy, float* uz, const float* vortx,
                  const float* vorty, const float* vortz, float coeff, float* lattice)
{
    const ptrdiff_t w = (ptrdiff_t)vort_first(g), zoff = (ptrdiff_t)g.z0 * g.plane;
    const VortView vx{vortx - w, g.nx, g.ny + 1, g.nz + 1}, vy{vorty - w, g.nx + 1, g.ny, g.nz + 1},
        vz{vortz - w, g.nx + 1, g.ny + 1, g.nz};
    dim3 block(64, 4);
    if (lattice) {
        int kb, ke;
        lattice_planes(g, &kb, &ke);
        Lattice Lt;
        Lt.sx = g.nx + 1;
        Lt.sxy = (g.nx + 1) * (g.ny + 1);
        Lt.k0 = kb;
        const size_t n = (size_t)Lt.sxy * (ke - kb);
        Lt.A = lattice; Lt.B = lattice + n; Lt.C = lattice + 2 * n; Lt.D = lattice + 3 * n;
        // faces with k in [3, nz - 2] read lattice planes k - 1 .. k + 1 (k + 1: ux and uy faces only)
        const int lb = max(kb, 2), le = min(ke, g.nz);
        if (le > lb)
            hipLaunchKernelGGL(g.fmad ? k_vort_lattice<true> : k_vort_lattice<false>,
                               dim3(cdiv(g.nx + 1, 64), cdiv(g.ny + 1, 4), le - lb), block, 0, st, Lt, vx, vy, vz, g.nx, g.ny, lb);
        const int fb = max(g.z0, 3), fe = min(g.z0 + g.nzl + 1, g.nz - 1);
        if (fe > fb)
            hipLaunchKernelGGL(g.fmad ? k_vort_confine_all<true> : k_vort_confine_all<false>,
                               dim3(cdiv(g.nx + 1, 64), cdiv(g.ny + 1, 4), fe - fb), block, 0, st,
                               ux - zoff, uy - zoff, uz - zoff, Lt, vx, vy, vz, g.nx, g.ny, g.nz, g.pitch, g.plane, coeff,
                               fb, g.z0 + g.nzl);
        return;
    }
    {
        const int dx = g.nx + 1, dy = g.ny, dz = g.nz;
        hipLaunchKernelGGL((g.fmad ? k_vort_confine<1, true> : k_vort_confine<1, false>), dim3(cdiv(dx, 64), cdiv(dy, 4), g.nzl), block, 0, st, ux - zoff, vx, vy,
                           vz, dx, dy, dz, g.pitch, g.plane, coeff, g.z0);
    }
    {
        const int dx = g.nx, dy = g.ny + 1, dz = g.nz;
        hipLaunchKernelGGL((g.fmad ? k_vort_confine<2, true> : k_vort_confine<2, false>), dim3(cdiv(dx, 64), cdiv(dy, 4), g.nzl), block, 0, st, uy - zoff, vx, vy,
                           vz, dx, dy, dz, g.pitch, g.plane, coeff, g.z0);
    }
    {   // the face plane shared with the upper neighbour is produced by both ranks
        const int dx = g.nx, dy = g.ny, dz = g.nz + 1;
        hipLaunchKernelGGL((g.fmad ? k_vort_confine<3, true> : k_vort_confine<3, false>), dim3(cdiv(dx, 64), cdiv(dy, 4), g.nzl + 1), block, 0, st, uz - zoff, vx,
                           vy, vz, dx, dy, dz, g.pitch, g.plane, coeff, g.z0);
    }
}

}  // namespace fsk
