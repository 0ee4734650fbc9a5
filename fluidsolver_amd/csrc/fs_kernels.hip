// fs_kernels.hip -- gfx950 (CDNA4, wave64) kernels of the smoke-solver step.
//
// Arithmetic contract: every field value is computed in IEEE binary32 in exactly the
// association order of the reference expressions, with no fused multiply-add (this file
// is compiled with -ffp-contract=off and the pragma below), so results are bit-identical
// to an uncontracted host evaluation of the same expressions up to the sign of zero.
//
// Reference expressions restated here (all in CuSolver/Source/Solver.cu):
//   zero-outside / clamped point samples :30-42, trilinear gather :52-113,
//   RK4 / MacCormack back-trace :139-169, source :295-307, buoyancy :346-368,
//   semi-Lagrangian gather :370-403, divergence :406-423, Jacobi :426-443,
//   pressure gradient :446-483, residual :442.
#include "fs_kernels.h"

#include <climits>

#pragma clang fp contract(off)

namespace {

constexpr int kWave = 64;

__host__ __device__ inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------- sampling helpers

struct Axis {
    int i;     // lower tap index
    float f;   // interpolation weight of the upper tap
};

// Integer/fraction split of one gather coordinate, Solver.cu:58-82.  The reference
// compares the float against the double literal -1e-6; no float lies between that
// literal and its float rounding (-1e-6f is the nearer-to-zero neighbour), so the float
// comparison below decides identically for every float input.
__device__ __forceinline__ Axis split_axis(float x)
{
    int ix = (int)x;                 // truncation toward zero
    float fx = x - floorf(x);
    if (x < -1e-6f) {
        ix -= 1;
        fx = x - (float)ix;
    }
    return Axis{ix, fx};
}

// Math.cuh:57-60
__device__ __forceinline__ float lerp1(float a, float b, float s)
{
    return a * (1.0f - s) + b * s;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi)
{
    return min(max(v, lo), hi);   // v_med3_i32
}

// Limits of the lower tap index along z for a gather on this slab.
struct ZRange {
    int lo, hi;          // clamp range of the lower tap
    int chk_lo, chk_hi;  // a lower tap outside [chk_lo, chk_hi] left the ghost planes
};

// Zero-outside trilinear gather (Solver.cu:52-113 with b_minmax_sampe = false) on the
// padded layout: clamping the lower tap to [-2, extent] lands every outside tap on a
// stored zero.  f is pre-offset so that global z indexes it directly.
template <bool HALO>
__device__ __forceinline__ float gather0(const float* __restrict__ f, Axis ax, Axis ay, Axis az,
                                         int dx, int dy, const ZRange& zr, int pitch, int plane,
                                         int& bad)
{
    if (HALO) bad |= (az.i < zr.chk_lo) | (az.i > zr.chk_hi);
    int ix = clampi(ax.i, -2, dx);
    int iy = clampi(ay.i, -2, dy);
    int iz = clampi(az.i, zr.lo, zr.hi);
    const float* q = f + (ix + iy * pitch + iz * plane);
    float a = q[0];
    float b = q[1];
    float c = q[pitch];
    float d = q[pitch + 1];
    const float* q2 = q + plane;
    float e = q2[0];
    float ff = q2[1];
    float g = q2[pitch];
    float h = q2[pitch + 1];
    float l1 = lerp1(lerp1(a, b, ax.f), lerp1(c, d, ax.f), ay.f);
    float l2 = lerp1(lerp1(e, ff, ax.f), lerp1(g, h, ax.f), ay.f);
    return lerp1(l1, l2, az.f);
}

struct Vel {
    const float* __restrict__ ux;
    const float* __restrict__ uy;
    const float* __restrict__ uz;
    int nx, ny;
    int pitch, plane;
    ZRange zc;   // z limits for cell-centred-in-z arrays (ux, uy: dz = nz)
    ZRange zf;   // z limits for uz (dz = nz + 1)
};

// Velocity at a point: the three staggered gathers of one RK stage (Solver.cu:142-144).
template <bool HALO>
__device__ __forceinline__ float3 velocity_at(const Vel& v, float3 p, int& bad)
{
    Axis x0 = split_axis(p.x - 0.f), xh = split_axis(p.x - 0.5f);
    Axis y0 = split_axis(p.y - 0.f), yh = split_axis(p.y - 0.5f);
    Axis z0 = split_axis(p.z - 0.f), zh = split_axis(p.z - 0.5f);
    float3 u;
    u.x = gather0<HALO>(v.ux, x0, yh, zh, v.nx + 1, v.ny, v.zc, v.pitch, v.plane, bad);
    u.y = gather0<HALO>(v.uy, xh, y0, zh, v.nx, v.ny + 1, v.zc, v.pitch, v.plane, bad);
    u.z = gather0<HALO>(v.uz, xh, yh, z0, v.nx, v.ny, v.zf, v.pitch, v.plane, bad);
    return u;
}

// Solver.cu:139-161.  (0.5f*dt) is formed first, then scales each component
// (Math.cuh:80-86); the final update is  pos - (dt * sum) / 6  with a true division.
template <bool HALO>
__device__ __forceinline__ float3 rk4(const Vel& v, float3 pos, float dt, int& bad)
{
    float hdt = 0.5f * dt;
    float3 u1 = velocity_at<HALO>(v, pos, bad);
    float3 p1 = make_float3(pos.x - hdt * u1.x, pos.y - hdt * u1.y, pos.z - hdt * u1.z);
    float3 u2 = velocity_at<HALO>(v, p1, bad);
    float3 p2 = make_float3(pos.x - hdt * u2.x, pos.y - hdt * u2.y, pos.z - hdt * u2.z);
    float3 u3 = velocity_at<HALO>(v, p2, bad);
    float3 p3 = make_float3(pos.x - dt * u3.x, pos.y - dt * u3.y, pos.z - dt * u3.z);
    float3 u4 = velocity_at<HALO>(v, p3, bad);
    float sx = ((u1.x + 2.0f * u2.x) + 2.0f * u3.x) + u4.x;
    float sy = ((u1.y + 2.0f * u2.y) + 2.0f * u3.y) + u4.y;
    float sz = ((u1.z + 2.0f * u2.z) + 2.0f * u3.z) + u4.z;
    return make_float3(pos.x - (dt * sx) / 6.0f, pos.y - (dt * sy) / 6.0f,
                       pos.z - (dt * sz) / 6.0f);
}

// Solver.cu:163-169 (MAC) or the plain RK4 trace of Solver.cu:396.
template <bool MAC, bool HALO>
__device__ __forceinline__ float3 backtrace(const Vel& v, float3 pos, float dt, int& bad)
{
    float3 pred = rk4<HALO>(v, pos, dt, bad);
    if (!MAC) return pred;
    float3 refl = rk4<HALO>(v, pred, -dt, bad);
    return make_float3(pred.x + (pos.x - refl.x) / 2.f, pred.y + (pos.y - refl.y) / 2.f,
                       pred.z + (pos.z - refl.z) / 2.f);
}

__host__ inline ZRange make_zrange(const FsGrid& g, int dz)
{
    ZRange r;
    r.lo = g.wall_lo ? -2 : g.z0 - g.ghost;
    r.hi = g.wall_hi ? dz : g.z0 + g.nzl + g.ghost - 1;
    r.chk_lo = g.wall_lo ? INT_MIN : r.lo;
    r.chk_hi = g.wall_hi ? INT_MAX : r.hi;
    return r;
}

__host__ inline Vel make_vel(const FsGrid& g, const float* ux, const float* uy, const float* uz)
{
    Vel v;
    ptrdiff_t zoff = (ptrdiff_t)g.z0 * g.plane;
    v.ux = ux - zoff;
    v.uy = uy - zoff;
    v.uz = uz - zoff;
    v.nx = g.nx;
    v.ny = g.ny;
    v.pitch = g.pitch;
    v.plane = g.plane;
    v.zc = make_zrange(g, g.nz);
    v.zf = make_zrange(g, g.nz + 1);
    return v;
}

// ---------------------------------------------------------------- small kernels

__global__ void k_fill_interior(float* __restrict__ f, float value, int nx, int ny, int pitch,
                                int plane)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y * blockDim.y + threadIdx.y;
    int z = blockIdx.z;
    if (x < nx && y < ny) f[x + y * pitch + z * plane] = value;
}

// Solver.cu:295-307, launched only over the bounding box of the sphere (cells outside it
// cannot satisfy the predicate).  zg indexes rho/T by global z.
__global__ void k_source(float* __restrict__ rho, float* __restrict__ T, FsSource s, int x0, int y0,
                         int z0, int x1, int y1, int z1, int pitch, int plane)
{
    int x = x0 + blockIdx.x * blockDim.x + threadIdx.x;
    int y = y0 + blockIdx.y * blockDim.y + threadIdx.y;
    int z = z0 + blockIdx.z;
    if (x > x1 || y > y1 || z > z1) return;
    float ddx = (float)x - s.x, ddy = (float)y - s.y, ddz = (float)z - s.z;
    // pow(v, 2) of the reference: the squares of these exactly representable
    // differences are exact, so v*v is the same number.
    if (ddx * ddx + ddy * ddy + ddz * ddz <= s.radius * s.radius) {
        int o = x + y * pitch + z * plane;
        rho[o] = s.rho0;
        T[o] = s.T0;
    }
}

// Solver.cu:346-368 (BURST off).  The gather point (i+.5, j, k+.5) has x and z weights
// exactly 0 and y weight exactly .5, so for finite T the clamped trilinear gather equals
// T(i,clamp(j-1),k)*.5 + T(i,clamp(j),k)*.5 with the same roundings (a*1 + b*0 == a).
__global__ void k_force(float* __restrict__ uy, const float* __restrict__ T, float dt, float T_env,
                        int nx, int ny, int pitch, int plane)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int j = blockIdx.y * blockDim.y + threadIdx.y;
    int z = blockIdx.z;
    if (x >= nx || j > ny) return;
    int jm = max(j - 1, 0);
    int jp = min(j, ny - 1);
    int zo = z * plane + x;
    float a = T[zo + jm * pitch];
    float c = T[zo + jp * pitch];
    float t = a * (1.0f - 0.5f) + c * 0.5f;
    float buoyancy = (t - T_env) * dt;
    uy[zo + j * pitch] += buoyancy;
}

// ---------------------------------------------------------------- advection

// One thread per output element; threads of a block walk the (x, y) extent of one z plane
// in flattened order so that fields of odd width (ux: nx+1) still fill every lane.
template <int DIR, bool MAC, bool HALO>
__global__ void __launch_bounds__(256)
k_advect_vel(const float* __restrict__ field, float* __restrict__ out, Vel v, float dt, int zfirst,
             int* halo_flag)
{
    const int dx = DIR == 1 ? v.nx + 1 : v.nx;
    const int dy = DIR == 2 ? v.ny + 1 : v.ny;
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= dx * dy) return;
    int j = t / dx;
    int i = t - j * dx;
    int k = zfirst + blockIdx.y;
    float3 pos;   // Solver.cu:381-383
    pos.x = (float)i + (DIR == 1 ? 0.f : 0.5f);
    pos.y = (float)j + (DIR == 2 ? 0.f : 0.5f);
    pos.z = (float)k + (DIR == 3 ? 0.f : 0.5f);
    int bad = 0;
    float3 c = backtrace<MAC, HALO>(v, pos, dt, bad);
    Axis ax = split_axis(c.x - (DIR == 1 ? 0.f : 0.5f));
    Axis ay = split_axis(c.y - (DIR == 2 ? 0.f : 0.5f));
    Axis az = split_axis(c.z - (DIR == 3 ? 0.f : 0.5f));
    float r = gather0<HALO>(field, ax, ay, az, dx, dy, DIR == 3 ? v.zf : v.zc, v.pitch, v.plane, bad);
    out[i + j * v.pitch + k * v.plane] = r;
    if (HALO && bad) atomicOr(halo_flag, 1);
}

template <bool MAC, bool HALO>
__global__ void __launch_bounds__(256)
k_advect_scalars(const float* __restrict__ T, float* __restrict__ nT, const float* __restrict__ rho,
                 float* __restrict__ nrho, Vel v, float dt, int zfirst, int* halo_flag)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= v.nx * v.ny) return;
    int j = t / v.nx;
    int i = t - j * v.nx;
    int k = zfirst + blockIdx.y;
    float3 pos = make_float3((float)i + 0.5f, (float)j + 0.5f, (float)k + 0.5f);
    int bad = 0;
    float3 c = backtrace<MAC, HALO>(v, pos, dt, bad);
    Axis ax = split_axis(c.x - 0.5f), ay = split_axis(c.y - 0.5f), az = split_axis(c.z - 0.5f);
    int o = i + j * v.pitch + k * v.plane;
    nT[o] = gather0<HALO>(T, ax, ay, az, v.nx, v.ny, v.zc, v.pitch, v.plane, bad);
    nrho[o] = gather0<HALO>(rho, ax, ay, az, v.nx, v.ny, v.zc, v.pitch, v.plane, bad);
    if (HALO && bad) atomicOr(halo_flag, 1);
}

// ---------------------------------------------------------------- projection

// Solver.cu:406-423.  Every read is in range, so the clamped sample never clamps.
__global__ void k_divergence(float* __restrict__ div, const float* __restrict__ ux,
                             const float* __restrict__ uy, const float* __restrict__ uz, int nx,
                             int ny, int pitch, int plane)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y * blockDim.y + threadIdx.y;
    int z = blockIdx.z;
    if (x >= nx || y >= ny) return;
    int o = x + y * pitch + z * plane;
    float ul = ux[o], ur = ux[o + 1];
    float ubo = uy[o], ut = uy[o + pitch];
    float ubh = uz[o], uf = uz[o + plane];
    div[o] = ur + uf + ut - ul - ubh - ubo;
}

// Solver.cu:446-483, the three kernels fused: thread (i,j,k) owns face i of ux, face j of
// uy and face k of uz where those exist.  Wall faces see two equal clamped samples and
// subtract exactly 0.
__global__ void k_gradient(float* __restrict__ ux, float* __restrict__ uy, float* __restrict__ uz,
                           const float* __restrict__ p, int nx, int ny, int nz, int pitch, int plane,
                           int zb, int ze_cells)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int j = blockIdx.y * blockDim.y + threadIdx.y;
    int k = zb + blockIdx.z;
    if (i > nx || j > ny) return;
    int ic = min(i, nx - 1), jc = min(j, ny - 1), kc = min(k, nz - 1);
    int oc = ic + jc * pitch + kc * plane;
    float pc = p[oc];
    int o = i + j * pitch + k * plane;
    if (j < ny && k < ze_cells) {
        float pl = p[max(i - 1, 0) + jc * pitch + kc * plane];
        ux[o] -= pc - pl;
    }
    if (i < nx && k < ze_cells) {
        float pbo = p[ic + max(j - 1, 0) * pitch + kc * plane];
        uy[o] -= pc - pbo;
    }
    if (i < nx && j < ny) {
        float pbh = p[ic + jc * pitch + max(k - 1, 0) * plane];
        uz[o] -= pc - pbh;
    }
}

// ---------------------------------------------------------------- residual

constexpr int kResBlock = 256;

__device__ __forceinline__ double block_sum(double v, double* sh)
{
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0)
        for (int q = 0; q < kResBlock / kWave; ++q) s += sh[q];
    return s;
}

// r = div + 6p - pl - pr - pbh - pf - pbo - pt in binary32 (Solver.cu:442), squared and
// summed in binary64; one partial per block, fixed order => reproducible.
__global__ void __launch_bounds__(kResBlock)
k_residual_partial(const float* __restrict__ p, const float* __restrict__ div, double* partial,
                   int nx, int ny, int nz, int pitch, int plane, int zb)
{
    __shared__ double sh[kResBlock / kWave];
    int y = blockIdx.x;
    int z = zb + blockIdx.y;
    double acc = 0.0;
    for (int x = threadIdx.x; x < nx; x += kResBlock) {
        int o = x + y * pitch + z * plane;
        float pc = p[o];
        float pl = p[x > 0 ? o - 1 : o];
        float pr = p[x < nx - 1 ? o + 1 : o];
        float pbo = p[y > 0 ? o - pitch : o];
        float pt = p[y < ny - 1 ? o + pitch : o];
        float pbh = p[z > 0 ? o - plane : o];
        float pf = p[z < nz - 1 ? o + plane : o];
        float r = div[o] + 6 * pc - pl - pr - pbh - pf - pbo - pt;
        acc += (double)r * (double)r;
    }
    double s = block_sum(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x + blockIdx.y * gridDim.x] = s;
}

__global__ void __launch_bounds__(kResBlock)
k_residual_final(const double* partial, int n, double* out)
{
    __shared__ double sh[kResBlock / kWave];
    double acc = 0.0;
    for (int q = threadIdx.x; q < n; q += kResBlock) acc += partial[q];
    double s = block_sum(acc, sh);
    if (threadIdx.x == 0) *out = s;
}

inline dim3 tile_grid(int dx, int dy, int dz) { return dim3(cdiv(dx, 64), cdiv(dy, 4), dz); }
const dim3 kTile(64, 4, 1);

}  // namespace

namespace fsk {

void fill_interior(hipStream_t st, const FsGrid& g, float* f, float value)
{
    hipLaunchKernelGGL(k_fill_interior, tile_grid(g.nx, g.ny, g.nzl), kTile, 0, st, f, value, g.nx,
                       g.ny, g.pitch, g.plane);
}

void source(hipStream_t st, const FsGrid& g, const FsSource& src, float* rho, float* T)
{
    // integer bounding box of the sphere, intersected with the owned cells
    int x0 = max(0, (int)floorf(src.x - src.radius)), x1 = min(g.nx - 1, (int)ceilf(src.x + src.radius));
    int y0 = max(0, (int)floorf(src.y - src.radius)), y1 = min(g.ny - 1, (int)ceilf(src.y + src.radius));
    int z0 = max(g.z0, (int)floorf(src.z - src.radius));
    int z1 = min(g.z0 + g.nzl - 1, (int)ceilf(src.z + src.radius));
    if (x1 < x0 || y1 < y0 || z1 < z0) return;
    ptrdiff_t zoff = (ptrdiff_t)g.z0 * g.plane;
    dim3 grid(cdiv(x1 - x0 + 1, 64), cdiv(y1 - y0 + 1, 4), z1 - z0 + 1);
    hipLaunchKernelGGL(k_source, grid, kTile, 0, st, rho - zoff, T - zoff, src, x0, y0, z0, x1, y1,
                       z1, g.pitch, g.plane);
}

void force_uy(hipStream_t st, const FsGrid& g, float* uy, const float* T, float dt, float T_env)
{
    hipLaunchKernelGGL(k_force, tile_grid(g.nx, g.ny + 1, g.nzl), kTile, 0, st, uy, T, dt, T_env,
                       g.nx, g.ny, g.pitch, g.plane);
}

template <int DIR>
static void launch_advect_vel(hipStream_t st, const FsGrid& g, const float* field, float* out,
                              const Vel& v, float dt, int scheme, bool halo, int* halo_flag)
{
    int dx = DIR == 1 ? g.nx + 1 : g.nx, dy = DIR == 2 ? g.ny + 1 : g.ny;
    int planes = DIR == 3 ? g.nzl + 1 : g.nzl;
    dim3 grid(cdiv(dx * dy, 256), planes), block(256);
    ptrdiff_t zoff = (ptrdiff_t)g.z0 * g.plane;
    const float* f = field - zoff;
    float* o = out - zoff;
    bool mac = scheme != 0;
#define FS_LAUNCH(M, H) \
    hipLaunchKernelGGL((k_advect_vel<DIR, M, H>), grid, block, 0, st, f, o, v, dt, g.z0, halo_flag)
    if (mac && halo) FS_LAUNCH(true, true);
    else if (mac) FS_LAUNCH(true, false);
    else if (halo) FS_LAUNCH(false, true);
    else FS_LAUNCH(false, false);
#undef FS_LAUNCH
}

void advect_velocity(hipStream_t st, const FsGrid& g, int dir, const float* field, float* out,
                     const float* ux, const float* uy, const float* uz, float dt, int scheme,
                     int* halo_flag)
{
    Vel v = make_vel(g, ux, uy, uz);
    bool halo = !(g.wall_lo && g.wall_hi);
    if (dir == 1) launch_advect_vel<1>(st, g, field, out, v, dt, scheme, halo, halo_flag);
    else if (dir == 2) launch_advect_vel<2>(st, g, field, out, v, dt, scheme, halo, halo_flag);
    else launch_advect_vel<3>(st, g, field, out, v, dt, scheme, halo, halo_flag);
}

void advect_scalars(hipStream_t st, const FsGrid& g, const float* T, float* nT, const float* rho,
                    float* nrho, const float* ux, const float* uy, const float* uz, float dt,
                    int scheme, int* halo_flag)
{
    Vel v = make_vel(g, ux, uy, uz);
    bool halo = !(g.wall_lo && g.wall_hi);
    bool mac = scheme != 0;
    ptrdiff_t zoff = (ptrdiff_t)g.z0 * g.plane;
    dim3 grid(cdiv(g.nx * g.ny, 256), g.nzl), block(256);
#define FS_LAUNCH(M, H)                                                                       \
    hipLaunchKernelGGL((k_advect_scalars<M, H>), grid, block, 0, st, T - zoff, nT - zoff,     \
                       rho - zoff, nrho - zoff, v, dt, g.z0, halo_flag)
    if (mac && halo) FS_LAUNCH(true, true);
    else if (mac) FS_LAUNCH(true, false);
    else if (halo) FS_LAUNCH(false, true);
    else FS_LAUNCH(false, false);
#undef FS_LAUNCH
}

void divergence(hipStream_t st, const FsGrid& g, float* div, const float* ux, const float* uy,
                const float* uz)
{
    hipLaunchKernelGGL(k_divergence, tile_grid(g.nx, g.ny, g.nzl), kTile, 0, st, div, ux, uy, uz,
                       g.nx, g.ny, g.pitch, g.plane);
}

void gradient(hipStream_t st, const FsGrid& g, float* ux, float* uy, float* uz, const float* p)
{
    ptrdiff_t zoff = (ptrdiff_t)g.z0 * g.plane;
    hipLaunchKernelGGL(k_gradient, tile_grid(g.nx + 1, g.ny + 1, g.nzl + 1), kTile, 0, st,
                       ux - zoff, uy - zoff, uz - zoff, p - zoff, g.nx, g.ny, g.nz, g.pitch, g.plane,
                       g.z0, g.z0 + g.nzl);
}

int residual_partials(const FsGrid& g) { return g.ny * g.nzl; }

void residual(hipStream_t st, const FsGrid& g, const float* p, const float* div, double* partial,
              double* out)
{
    ptrdiff_t zoff = (ptrdiff_t)g.z0 * g.plane;
    hipLaunchKernelGGL(k_residual_partial, dim3(g.ny, g.nzl), dim3(kResBlock), 0, st, p - zoff,
                       div - zoff, partial, g.nx, g.ny, g.nz, g.pitch, g.plane, g.z0);
    hipLaunchKernelGGL(k_residual_final, dim3(1), dim3(kResBlock), 0, st, partial,
                       residual_partials(g), out);
}

}  // namespace fsk
