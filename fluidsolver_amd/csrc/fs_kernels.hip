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

#include <cstdint>
#include <cstdlib>
#include <mutex>

#pragma clang fp contract(off)

namespace {

constexpr int kWave = 64;

__host__ __device__ inline int cdiv(int a, int b) { return (a + b - 1) / b; }

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
// FMAD: lerp's left product and the final multiply-add fused (FS_FLAG_FMAD).
template <bool FMAD>
__device__ __forceinline__ float buoyant(float uy, float a, float c, float dt, float T_env)
{
    if constexpr (FMAD) return __builtin_fmaf(__builtin_fmaf(a, 1.0f - 0.5f, c * 0.5f) - T_env, dt, uy);
    else return uy + (a * (1.0f - 0.5f) + c * 0.5f - T_env) * dt;
}

template <bool FMAD>
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
    uy[zo + j * pitch] = buoyant<FMAD>(uy[zo + j * pitch], a, c, dt, T_env);
}

// the same for four faces along x per thread (nx % 4 == 0; rows are 64-byte aligned)
template <bool FMAD>
__global__ void k_force4(float* __restrict__ uy, const float* __restrict__ T, float dt, float T_env,
                         int nx4, int ny, int pitch, int plane)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    int x4 = blockIdx.x * blockDim.x + threadIdx.x;
    int j = blockIdx.y * blockDim.y + threadIdx.y;
    int z = blockIdx.z;
    if (x4 >= nx4 || j > ny) return;
    int jm = max(j - 1, 0);
    int jp = min(j, ny - 1);
    int zo = z * plane + 4 * x4;
    const f4 a = *(const f4*)(T + zo + jm * pitch);
    const f4 c = *(const f4*)(T + zo + jp * pitch);
    f4* dst = (f4*)(uy + zo + j * pitch);
    f4 u = *dst;
    u.x = buoyant<FMAD>(u.x, a.x, c.x, dt, T_env);
    u.y = buoyant<FMAD>(u.y, a.y, c.y, dt, T_env);
    u.z = buoyant<FMAD>(u.z, a.z, c.z, dt, T_env);
    u.w = buoyant<FMAD>(u.w, a.w, c.w, dt, T_env);
    *dst = u;
}

// Up to kCopyBatchMax contiguous float ranges copied by one launch (blockIdx.y = range).  Used to
// pack halo planes into, and unpack them from, pinned host memory that the GPU addresses directly.
__global__ void __launch_bounds__(256) k_copy_batch(fsk::CopyBatch b)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int r = blockIdx.y;
    const size_t n = b.n[r];
    float* __restrict__ dst = b.dst[r];
    const float* __restrict__ src = b.src[r];
    const size_t stride = (size_t)gridDim.x * blockDim.x, t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if ((((uintptr_t)dst | (uintptr_t)src) & 15u) == 0) {
        const size_t n4 = n / 4;
        for (size_t i = t; i < n4; i += stride) ((f4*)dst)[i] = ((const f4*)src)[i];
        for (size_t i = 4 * n4 + t; i < n; i += stride) dst[i] = src[i];
    } else {
        for (size_t i = t; i < n; i += stride) dst[i] = src[i];
    }
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

// The same with four cells along x per thread (nx % 4 == 0): thread (x4, j, k) owns faces
// 4*x4 .. 4*x4+3 of ux, uy, uz; the thread of the last quad also owns the wall face i = nx of ux.
// Every face evaluates the expression of k_gradient on the same clamped samples.
__global__ void k_gradient4(float* __restrict__ ux, float* __restrict__ uy, float* __restrict__ uz,
                            const float* __restrict__ p, int nx, int ny, int nz, int pitch, int plane,
                            int zb, int ze_cells)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int x4 = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y * blockDim.y + threadIdx.y;
    const int k = zb + blockIdx.z;
    const int i = 4 * x4;
    if (i >= nx || j > ny) return;
    const int jc = min(j, ny - 1), kc = min(k, nz - 1);
    const f4 pc = *(const f4*)(p + i + jc * pitch + kc * plane);
    const int o = i + j * pitch + k * plane;
    if (j < ny && k < ze_cells) {
        const float pl = p[max(i - 1, 0) + jc * pitch + kc * plane];
        f4 u = *(f4*)(ux + o);
        u.x -= pc.x - pl;
        u.y -= pc.y - pc.x;
        u.z -= pc.z - pc.y;
        u.w -= pc.w - pc.z;
        *(f4*)(ux + o) = u;
        if (i + 4 == nx) ux[o + 4] -= pc.w - pc.w;   // face i = nx: both clamped samples are p[nx-1]
    }
    if (k < ze_cells) {
        const f4 pbo = *(const f4*)(p + i + max(j - 1, 0) * pitch + kc * plane);
        f4 u = *(f4*)(uy + o);
        u -= pc - pbo;
        *(f4*)(uy + o) = u;
    }
    if (j < ny) {
        const f4 pbh = *(const f4*)(p + i + jc * pitch + max(k - 1, 0) * plane);
        f4 u = *(f4*)(uz + o);
        u -= pc - pbh;
        *(f4*)(uz + o) = u;
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

// Scratch of the residual sum for callers that must not allocate (fs_residual_noalloc): one partial
// per (y, z) row, then the total.  Every GPU this library runs on holds its own copy of the array
// (8 MiB) until the process ends.  In host code the name below is only a handle for
// hipGetSymbolAddress: kernels and copies take the per-device address that call returns.
constexpr int kStaticRows = 1 << 20;
static __device__ __align__(256) double scratch[kStaticRows + 32];
constexpr int kMaxDevices = 64;
double* g_scratch_ptr[kMaxDevices];   // device address of `scratch` on each device, null = not looked up
std::mutex g_scratch_mu;

inline dim3 tile_grid(int dx, int dy, int dz) { return dim3(cdiv(dx, 64), cdiv(dy, 4), dz); }
const dim3 kTile(64, 4, 1);

}  // namespace

namespace fsk {

void copy_batch(hipStream_t st, const CopyBatch& b)
{
    if (b.count <= 0) return;
    size_t nmax = 0;
    for (int r = 0; r < b.count; ++r) nmax = b.n[r] > nmax ? b.n[r] : nmax;
    if (nmax == 0) return;
    const unsigned blocks = (unsigned)((nmax / 4 + 255) / 256);
    hipLaunchKernelGGL(k_copy_batch, dim3(blocks < 1 ? 1 : (blocks > 512 ? 512 : blocks), b.count), dim3(256), 0, st, b);
}

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
    if (g.nx % 4 == 0) {
        if (g.fmad) hipLaunchKernelGGL(k_force4<true>, tile_grid(g.nx / 4, g.ny + 1, g.nzl), kTile, 0, st, uy, T, dt,
                                       T_env, g.nx / 4, g.ny, g.pitch, g.plane);
        else hipLaunchKernelGGL(k_force4<false>, tile_grid(g.nx / 4, g.ny + 1, g.nzl), kTile, 0, st, uy, T, dt, T_env,
                                g.nx / 4, g.ny, g.pitch, g.plane);
    } else {
        if (g.fmad) hipLaunchKernelGGL(k_force<true>, tile_grid(g.nx, g.ny + 1, g.nzl), kTile, 0, st, uy, T, dt, T_env,
                                       g.nx, g.ny, g.pitch, g.plane);
        else hipLaunchKernelGGL(k_force<false>, tile_grid(g.nx, g.ny + 1, g.nzl), kTile, 0, st, uy, T, dt, T_env,
                                g.nx, g.ny, g.pitch, g.plane);
    }
}

void divergence(hipStream_t st, const FsGrid& g, float* div, const float* ux, const float* uy,
                const float* uz, int zb, int ze)
{
    if (ze <= zb) return;
    // planes [zb, ze) in global indices; local plane 0 is global z0
    ptrdiff_t o = (ptrdiff_t)(zb - g.z0) * g.plane;
    hipLaunchKernelGGL(k_divergence, tile_grid(g.nx, g.ny, ze - zb), kTile, 0, st, div + o, ux + o,
                       uy + o, uz + o, g.nx, g.ny, g.pitch, g.plane);
}

void gradient(hipStream_t st, const FsGrid& g, float* ux, float* uy, float* uz, const float* p)
{
    ptrdiff_t zoff = (ptrdiff_t)g.z0 * g.plane;
    if (g.nx % 4 == 0) {
        hipLaunchKernelGGL(k_gradient4, tile_grid(g.nx / 4, g.ny + 1, g.nzl + 1), kTile, 0, st,
                           ux - zoff, uy - zoff, uz - zoff, p - zoff, g.nx, g.ny, g.nz, g.pitch, g.plane,
                           g.z0, g.z0 + g.nzl);
        return;
    }
    hipLaunchKernelGGL(k_gradient, tile_grid(g.nx + 1, g.ny + 1, g.nzl + 1), kTile, 0, st,
                       ux - zoff, uy - zoff, uz - zoff, p - zoff, g.nx, g.ny, g.nz, g.pitch, g.plane,
                       g.z0, g.z0 + g.nzl);
}

int residual_partials(const FsGrid& g) { return g.ny * g.nzl; }

int residual_static_rows() { return kStaticRows; }

hipError_t residual_static_lookup(int device)
{
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (g_scratch_ptr[device]) return hipSuccess;
    int cur = -1;
    hipError_t e = hipGetDevice(&cur);
    if (e != hipSuccess) return e;
    if (cur != device) return hipErrorInvalidDevice;   // the address is the current device's copy
    double* ptr = nullptr;
    e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (!ptr) return hipErrorInvalidSymbol;
    g_scratch_ptr[device] = ptr;
    return hipSuccess;
}

bool residual_static_scratch(int device, double** partial, double** out)
{
    if (device < 0 || device >= kMaxDevices) return false;
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    double* ptr = g_scratch_ptr[device];
    if (!ptr) return false;
    *partial = ptr;
    *out = ptr + kStaticRows;
    return true;
}

void residual(hipStream_t st, const FsGrid& g, const float* p, const float* div, double* partial,
              double* out)
{
    ptrdiff_t zoff = (ptrdiff_t)g.z0 * g.plane;
    hipLaunchKernelGGL(k_residual_partial, dim3(g.ny, g.nzl), dim3(kResBlock), 0, st, p - zoff,
                       div - zoff, partial, g.nx, g.ny, g.nz, g.pitch, g.plane, g.z0);
    hipLaunchKernelGGL(k_residual_final, dim3(1), dim3(kResBlock), 0, st, partial,
                       residual_partials(g), out);
}

FsEnv read_env()
{
    FsEnv v;
    auto geti = [](const char* name, int& out) {
        if (const char* e = getenv(name)) out = atoi(e);
    };
    geti("FS_JACOBI_CFG", v.jacobi_cfg);
    geti("FS_JACOBI_NT", v.jacobi_nt);
    geti("FS_JACOBI_ORDER", v.jacobi_order);
    geti("FS_JACOBI_LZ", v.jacobi_lz);
    geti("FS_JACOBI_TUNE", v.jacobi_tune);
    geti("FS_JACOBI_DEEP", v.jacobi_deep);
    geti("FS_JACOBI_DEEP_LZ", v.jacobi_deep_lz);
    geti("FS_JACOBI_DEEP_NT", v.jacobi_deep_nt);
    geti("FS_ADVECT_GLOBAL", v.advect_global);
    geti("FS_ADVECT_BRICK", v.advect_brick);
    if (const char* e = getenv("FS_MG_TB_MIN")) v.mg_tb_min = atol(e);
    geti("FS_MG_LDS_PAD", v.mg_lds_pad);
    geti("FS_MG_NO_LDS", v.mg_no_lds);
    geti("FS_MG_SLAB_GHOST", v.mg_slab_ghost);
    geti("FS_VORT_LATTICE", v.vort_lattice);
    geti("FS_SLAB_PGHOST", v.slab_pghost);
    geti("FS_XCHG_STREAMS", v.xchg_streams);
    geti("FS_DEBUG_XCHG", v.xchg_debug);
    geti("FS_XCHG_HOST_WAIT", v.xchg_host_wait);
    geti("FS_XCHG_TIMEOUT_MS", v.xchg_timeout_ms);
    return v;
}

}  // namespace fsk
