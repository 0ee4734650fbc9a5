// fs_jacobi.hip -- temporally blocked Jacobi pressure sweeps for gfx950.
//
// One sweep is the reference's JacobiKernel (CuSolver/Source/Solver.cu:426-443):
//     p'(i,j,k) = (pl + pr + pbh + pf + pbo + pt - div) / 6      (that association, true divide)
// with neighbours clamped at the six global walls (sample(), Solver.cu:30-34).
//
// k_jacobi_tb applies T sweeps in ONE pass over memory ("3.5-D blocking"): a workgroup owns an
// (x, y) tile plus a halo, marches along z, and keeps T time levels in flight.  At march
// step n level 0 loads plane n, level 1 produces plane n-1, ... level T produces plane n-T,
// which is stored.  A thread owns one float4 column per tile row it is mapped to: its own
// z-neighbours (planes k-1, k+1 of the previous level) stay in registers, the x-neighbours come
// from the adjacent lanes of its 16-lane row through DPP row shifts, and only the y-neighbours go
// through LDS (one plane per level, double buffered => one barrier per march step).  Every cell
// value of every level is computed by the very expression above from the previous level, so the
// result is bit-identical to T separate sweeps; halo cells are recomputed redundantly.
//
// Algorithmic traffic stays 12 B per cell per sweep (read p, read div, write p'); the blocked
// pass moves about (1.4*4 + 1.3*4 + 4) B per cell per T sweeps through HBM.
#include "fs_kernels.h"

#include <cstdlib>

#pragma clang fp contract(off)

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// value of `v` held by the previous / next lane of the same 16-lane row (0 at the row ends)
__device__ __forceinline__ float row_prev(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
}
__device__ __forceinline__ float row_next(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, true));
}

constexpr int kW4 = 16;          // float4 columns per tile row (one 16-lane DPP row)
constexpr int kW = 4 * kW4;      // 64 cells per tile row
constexpr int kHaloX = 4;        // x halo is one float4, enough for T <= 4 and keeps loads aligned
constexpr int kBX = kW - 2 * kHaloX;

struct TbArgs {
    const float* p;      // pre-offset: global z indexes directly
    float* np;
    const float* div;
    int nx, ny, nz;
    int pitch, plane;
    int zin_lo, zin_hi;  // level-0 planes that may be read: [zin_lo, zin_hi)
    int zout_lo, zout_hi;// planes to produce
    int lz;              // output planes per workgroup along z
    int tiles_x, tiles_y, nblocks;
    int order;           // 0: plain block order, 1: XCD-contiguous (z-chunk, y, x) order
};

// Workgroups are dealt round-robin to the 8 XCDs, so ids b and b+8 share an L2.  Give each
// XCD one contiguous run of logical ids (bijective for any count): tiles that share halo rows
// and columns then run on one XCD at about the same time and meet in its L2.
__device__ __forceinline__ int xcd_contiguous(int b, int nb)
{
    int q = nb >> 3, r = nb & 7;
    int xcd = b & 7, loc = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

// x / 6 rounded to nearest, bit-identical to the IEEE division for every finite x whose
// quotient is a normal number: q = RN(x*c) with c = RN(1/6) is within one ulp of x/6, the
// remainder r = x - 6q is exact in an FMA, and RN(q + r*c) is then the correctly rounded
// quotient (x/6 is never within 2^-25 ulp of a rounding boundary because 6 = 2*3).  Zero,
// subnormal, infinite and NaN quotients take the true division.
__device__ __forceinline__ float div6_fast(float x)
{
    const float c = 0x1.555556p-3f;   // RN(1/6)
    float q = x * c;
    float r = __builtin_fmaf(-6.0f, q, x);
    return __builtin_fmaf(r, c, q);
}
// classes of q that div6_fast does not cover: NaNs, infinities, subnormals
__device__ __forceinline__ bool div6_needs_slow(float q)
{
    return __builtin_amdgcn_classf(q, 0x001 | 0x002 | 0x004 | 0x200 | 0x010 | 0x080);
}

// Per-thread registers of the march: for level t, a = plane n-t-2, b = plane n-t-1 and
// cc = plane n-t of that level; dv[t] = div of plane n-t; q_p / q_d = loads in flight, two
// march steps deep: slot n & 1 holds p of plane n and div of plane n-1 when step n starts and is
// refilled by step n with p of plane n+2 and div of plane n+1.
template <int T, int CPT>
struct TbRegs {
    f4 a[T][CPT], b[T][CPT], cc[T][CPT], dv[T + 1][CPT], q_p[2][CPT], q_d[2][CPT];
};

// Per-thread constants of the march.
template <int CPT>
struct TbLane {
    int ld_off[CPT];     // clamped into the domain: loads are unconditional and always legal
    int st_off[CPT];
    int row[CPT];        // own LDS row
    int row_dn[CPT];     // LDS row of the y-1 neighbour (own row at the y = 0 wall)
    int row_up[CPT];     // LDS row of the y+1 neighbour (own row at the y = ny-1 wall)
    bool store[CPT];
    bool at_left, at_right;
    int tx;
};

struct TbRange {
    int n_begin, n_end;  // level-0 planes this workgroup reads: [n_begin, n_end)
    int out_lo, out_hi;  // planes it stores
};

// One march step.  EDGE = false is the steady state: every level is active, no plane touched
// is a z wall, the next plane exists and the stored plane is in range, so the body is branch-free
// apart from the rare slow-division path.
template <int T, int NTY, int CPT, bool EDGE, int PH>
__device__ __forceinline__ void tb_step(int n, TbRegs<T, CPT>& R, const TbLane<CPT>& L,
                                        const TbArgs& A, const TbRange& Z, f4* lds)
{
    // PH == n & 1: selects the load slot and the LDS buffer statically
    constexpr int H = NTY * CPT;
    constexpr int cur = PH, prv = PH ^ 1;
    f4* lcur = lds + cur * (T * H * kW4) + L.tx;
    const f4* lprv = lds + prv * (T * H * kW4) + L.tx;

#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        R.cc[0][c] = R.q_p[PH][c];
        R.dv[1][c] = R.q_d[PH][c];
    }
    if (!EDGE || n + 2 < Z.n_end) {
#pragma unroll
        for (int c = 0; c < CPT; ++c)
            R.q_p[PH][c] = *reinterpret_cast<const f4*>(A.p + L.ld_off[c] + (n + 2) * A.plane);
    }
    if (!EDGE || n + 1 < Z.n_end) {
#pragma unroll
        for (int c = 0; c < CPT; ++c)
            R.q_d[PH][c] = *reinterpret_cast<const f4*>(A.div + L.ld_off[c] + (n + 1) * A.plane);
    }
    if (!EDGE || n < Z.n_end) {
#pragma unroll
        for (int c = 0; c < CPT; ++c) lcur[L.row[c] * kW4] = R.cc[0][c];
        if (EDGE && n == 0) {   // plane 0: the clamped z-1 neighbour is the plane itself
#pragma unroll
            for (int c = 0; c < CPT; ++c) R.b[0][c] = R.cc[0][c];
        }
    }
#pragma unroll
    for (int t = 1; t <= T; ++t) {
        const int k = n - t;                               // plane this level produces now
        if (EDGE && (k < Z.n_begin || k >= Z.n_end)) continue;   // uniform over the workgroup
        if (EDGE && k == A.nz - 1) {                       // clamped z+1 neighbour at the far wall
#pragma unroll
            for (int c = 0; c < CPT; ++c) R.cc[t - 1][c] = R.b[t - 1][c];
        }
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            const f4 ctr = R.b[t - 1][c];                  // previous level, plane k
            const f4 pbo = lprv[((t - 1) * H + L.row_dn[c]) * kW4];   // y - 1
            const f4 pt = lprv[((t - 1) * H + L.row_up[c]) * kW4];    // y + 1
            const f4 pbh = R.a[t - 1][c];                  // z - 1
            const f4 pf = R.cc[t - 1][c];                  // z + 1
            float lft = row_prev(ctr.w), rgt = row_next(ctr.x);
            if (L.at_left) lft = ctr.x;
            if (L.at_right) rgt = ctr.w;
            const f4 d = R.dv[t][c];
            const float s0 = lft + ctr.y + pbh.x + pf.x + pbo.x + pt.x - d.x;
            const float s1 = ctr.x + ctr.z + pbh.y + pf.y + pbo.y + pt.y - d.y;
            const float s2 = ctr.y + ctr.w + pbh.z + pf.z + pbo.z + pt.z - d.z;
            const float s3 = ctr.z + rgt + pbh.w + pf.w + pbo.w + pt.w - d.w;
            f4 r = {div6_fast(s0), div6_fast(s1), div6_fast(s2), div6_fast(s3)};
            if ((int)div6_needs_slow(r.x) | (int)div6_needs_slow(r.y) | (int)div6_needs_slow(r.z) |
                (int)div6_needs_slow(r.w)) {
                r.x = s0 / 6.f;
                r.y = s1 / 6.f;
                r.z = s2 / 6.f;
                r.w = s3 / 6.f;
            }
            if (t < T) {
                R.cc[t][c] = r;
                lcur[(t * H + L.row[c]) * kW4] = r;
            } else if (L.store[c] && (!EDGE || (k >= Z.out_lo && k < Z.out_hi))) {
                *reinterpret_cast<f4*>(A.np + L.st_off[c] + k * A.plane) = r;
            }
        }
        if (EDGE && t < T && k == 0) {   // plane 0 of level t: clamped z-1 neighbour for level t+1
#pragma unroll
            for (int c = 0; c < CPT; ++c) R.b[t][c] = R.cc[t][c];
        }
    }
    // slide the windows one plane
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            R.a[t][c] = R.b[t][c];
            R.b[t][c] = R.cc[t][c];
        }
#pragma unroll
    for (int t = T; t >= 2; --t)
#pragma unroll
        for (int c = 0; c < CPT; ++c) R.dv[t][c] = R.dv[t - 1][c];
    __syncthreads();
}

template <int T, int NTY, int CPT>
__global__ void __launch_bounds__(kW4 * NTY)
k_jacobi_tb(TbArgs A)
{
    constexpr int H = NTY * CPT;          // tile rows including the y halo
    constexpr int BY = H - 2 * T;
    static_assert(T >= 1 && T <= kHaloX, "x halo is one float4");
    static_assert(BY > 0, "tile too small");
    __shared__ f4 lds[2 * T * H * kW4];

    TbLane<CPT> L;
    L.tx = threadIdx.x;
    const int ty = threadIdx.y;
    // tile (bx, by) and z chunk bz of this workgroup
    const int lid = A.order ? xcd_contiguous((int)blockIdx.x, A.nblocks) : (int)blockIdx.x;
    const int bz = lid / (A.tiles_x * A.tiles_y);
    const int rem = lid - bz * (A.tiles_x * A.tiles_y);
    const int by = rem / A.tiles_x;
    const int bx = rem - by * A.tiles_x;
    const int x = bx * kBX - kHaloX + 4 * L.tx;                  // first cell of the column
    const bool x_in = x >= 0 && x < A.nx;                        // nx % 4 == 0: all four or none
    L.at_left = x == 0;
    L.at_right = x + 4 == A.nx;
    const bool x_out = L.tx >= 1 && L.tx < kW4 - 1;              // column is in the output tile
    const int xc = min(max(x, 0), A.nx - 4);
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        const int row = ty + c * NTY;
        const int y = by * BY - T + row;
        const bool in = x_in && y >= 0 && y < A.ny;
        L.row[c] = row;
        // clamped y-neighbours (Solver.cu:435-436): a wall row reads itself
        L.row_dn[c] = (y == 0 || row == 0) ? row : row - 1;
        L.row_up[c] = (y == A.ny - 1 || row == H - 1) ? row : row + 1;
        L.store[c] = in && x_out && row >= T && row < H - T;
        L.st_off[c] = x + y * A.pitch;
        L.ld_off[c] = xc + min(max(y, 0), A.ny - 1) * A.pitch;
    }

    TbRange Z;
    Z.out_lo = A.zout_lo + bz * A.lz;
    Z.out_hi = min(Z.out_lo + A.lz, A.zout_hi);
    Z.n_begin = max(Z.out_lo - T, A.zin_lo);
    Z.n_end = min(Z.out_hi + T, A.zin_hi);

    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    TbRegs<T, CPT> R;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int c = 0; c < CPT; ++c) R.a[t][c] = R.b[t][c] = R.cc[t][c] = zero;
#pragma unroll
    for (int t = 0; t <= T; ++t)
#pragma unroll
        for (int c = 0; c < CPT; ++c) R.dv[t][c] = zero;
    // prime the two load slots: p of planes n_begin and n_begin + 1, div of plane n_begin
    {
        const int s0 = Z.n_begin & 1;
        const bool two = Z.n_begin + 1 < Z.n_end;
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            const f4 p0 = *reinterpret_cast<const f4*>(A.p + L.ld_off[c] + Z.n_begin * A.plane);
            const f4 d0 = *reinterpret_cast<const f4*>(A.div + L.ld_off[c] + Z.n_begin * A.plane);
            f4 p1 = zero;
            if (two) p1 = *reinterpret_cast<const f4*>(A.p + L.ld_off[c] + (Z.n_begin + 1) * A.plane);
            R.q_p[0][c] = s0 ? p1 : p0;
            R.q_p[1][c] = s0 ? p0 : p1;
            R.q_d[0][c] = s0 ? d0 : zero;
            R.q_d[1][c] = s0 ? zero : d0;
        }
    }

    // steady range: all levels active (n-T >= n_begin), no z wall among planes n-T..n+2 and
    // plane n+2 still readable, stored plane n-T inside [out_lo, out_hi)
    const int steady_lo = max(Z.n_begin + T + 1, Z.out_lo + T);
    const int steady_hi = min(Z.n_end - 2, Z.out_hi + T);
    const int last = Z.n_end + T;
    int n = Z.n_begin;
    auto edge_step = [&](int m) {
        if (m & 1) tb_step<T, NTY, CPT, true, 1>(m, R, L, A, Z, lds);
        else tb_step<T, NTY, CPT, true, 0>(m, R, L, A, Z, lds);
    };
    for (; n < last && (n < steady_lo || (n & 1)); ++n) edge_step(n);
    for (; n + 1 < steady_hi; n += 2) {
        tb_step<T, NTY, CPT, false, 0>(n, R, L, A, Z, lds);
        tb_step<T, NTY, CPT, false, 1>(n + 1, R, L, A, Z, lds);
    }
    for (; n < last; ++n) edge_step(n);
}

// plain one-sweep kernel (any nx): one cell per thread
__global__ void k_jacobi_1(const float* __restrict__ p, float* __restrict__ np,
                           const float* __restrict__ div, int nx, int ny, int nz, int pitch, int plane,
                           int zb)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y * blockDim.y + threadIdx.y;
    int z = zb + blockIdx.z;
    if (x >= nx || y >= ny) return;
    int o = x + y * pitch + z * plane;
    float pl = p[x > 0 ? o - 1 : o];
    float pr = p[x < nx - 1 ? o + 1 : o];
    float pbo = p[y > 0 ? o - pitch : o];
    float pt = p[y < ny - 1 ? o + pitch : o];
    float pbh = p[z > 0 ? o - plane : o];
    float pf = p[z < nz - 1 ? o + plane : o];
    float d = div[o];
    np[o] = (pl + pr + pbh + pf + pbo + pt - d) / 6.f;
}

constexpr int kT = 4;        // sweeps fused per blocked pass

bool can_block(const FsGrid& g) { return g.nx % 4 == 0 && g.nx >= 8; }

// tile variants: rows of threads (NTY) x columns per thread (CPT); tile rows H = NTY*CPT
struct TbConfig { int nty, cpt; };
constexpr TbConfig kConfigs[] = {{40, 1}, {32, 1}, {64, 1}, {36, 2}, {24, 1}, {48, 1}};
constexpr int kNumConfigs = sizeof(kConfigs) / sizeof(kConfigs[0]);

int tb_config()
{
    static int cfg = [] {
        const char* e = getenv("FS_JACOBI_CFG");
        int v = e ? atoi(e) : 2;
        return (v >= 0 && v < kNumConfigs) ? v : 2;
    }();
    return cfg;
}

int tb_order()
{
    static int v = [] {
        const char* e = getenv("FS_JACOBI_ORDER");
        return e ? atoi(e) : 0;
    }();
    return v;
}

int tb_chunk_planes()
{
    static int v = [] {
        const char* e = getenv("FS_JACOBI_LZ");
        return e ? atoi(e) : 0;
    }();
    return v;
}

template <int T, int NTY, int CPT>
void launch_tb_cfg(hipStream_t st, const FsGrid& g, TbArgs A)
{
    constexpr int H = NTY * CPT, BY = H - 2 * T;
    int tiles = cdiv(g.nx, kBX) * cdiv(g.ny, BY);
    int nout = A.zout_hi - A.zout_lo;
    // Cut z into chunks so that the workgroups fill whole rounds of the chip.  A workgroup costs
    // (planes + 2T) march steps (T planes of ramp-up and drain at each chunk edge), and with
    // `slots` workgroups resident at once the launch takes ceil(blocks / slots) rounds.
    constexpr int lds_bytes = 2 * T * H * kW4 * 16;
    constexpr int per_cu_lds = 163840 / lds_bytes;
    constexpr int per_cu_waves = 16 / (kW4 * NTY / 64 > 16 ? 16 : kW4 * NTY / 64);   // <= 4 waves/SIMD at ~128 VGPRs
    const int slots = 256 * max(1, min(per_cu_lds, per_cu_waves));
    int chunks = 1;
    long best = -1;
    for (int c = 1; c <= max(1, nout / 8); ++c) {
        long rounds = cdiv(tiles * c, slots);
        long cost = rounds * (cdiv(nout, c) + 2 * T);
        // a launch that leaves CUs idle in its only round pays for the idle ones too
        if (tiles * c < slots) cost = cost * slots / (tiles * c);
        if (best < 0 || cost < best) {
            best = cost;
            chunks = c;
        }
    }
    A.lz = cdiv(nout, chunks);
    if (tb_chunk_planes() > 0) A.lz = min(nout, tb_chunk_planes());
    chunks = cdiv(nout, A.lz);
    A.tiles_x = cdiv(g.nx, kBX);
    A.tiles_y = cdiv(g.ny, BY);
    A.nblocks = A.tiles_x * A.tiles_y * chunks;
    A.order = tb_order();
    dim3 grid(A.nblocks), block(kW4, NTY);
    hipLaunchKernelGGL((k_jacobi_tb<T, NTY, CPT>), grid, block, 0, st, A);
}

template <int T>
void launch_tb(hipStream_t st, const FsGrid& g, const float* p, float* np, const float* div,
               int zin_lo, int zin_hi, int zout_lo, int zout_hi)
{
    TbArgs A;
    ptrdiff_t zoff = (ptrdiff_t)g.z0 * g.plane;
    A.p = p - zoff;
    A.np = np - zoff;
    A.div = div - zoff;
    A.nx = g.nx; A.ny = g.ny; A.nz = g.nz;
    A.pitch = g.pitch; A.plane = g.plane;
    A.zin_lo = zin_lo; A.zin_hi = zin_hi;
    A.zout_lo = zout_lo; A.zout_hi = zout_hi;
    A.lz = 0;
    switch (tb_config()) {
    case 1: launch_tb_cfg<T, 32, 1>(st, g, A); break;
    case 2: launch_tb_cfg<T, 64, 1>(st, g, A); break;
    case 3: launch_tb_cfg<T, 36, 2>(st, g, A); break;
    case 4: launch_tb_cfg<T, 24, 1>(st, g, A); break;
    case 5: launch_tb_cfg<T, 48, 1>(st, g, A); break;
    default: launch_tb_cfg<T, 40, 1>(st, g, A); break;
    }
}

}  // namespace

namespace fsk {

void jacobi_sweep(hipStream_t st, const FsGrid& g, const float* p, float* np, const float* div,
                  int zb, int ze)
{
    if (ze <= zb) return;
    ptrdiff_t zoff = (ptrdiff_t)g.z0 * g.plane;
    dim3 grid(cdiv(g.nx, 64), cdiv(g.ny, 4), ze - zb), block(64, 4);
    hipLaunchKernelGGL(k_jacobi_1, grid, block, 0, st, p - zoff, np - zoff, div - zoff, g.nx, g.ny,
                       g.nz, g.pitch, g.plane, zb);
}

int jacobi_block() { return kT; }

bool jacobi_pass(hipStream_t st, const FsGrid& g, const float* p, float* np, const float* div,
                 int sweeps, int zin_lo, int zin_hi, int zout_lo, int zout_hi)
{
    if (!can_block(g) || sweeps < 1 || sweeps > kT) return false;
    if (zout_hi <= zout_lo) return true;
    switch (sweeps) {
    case 1: launch_tb<1>(st, g, p, np, div, zin_lo, zin_hi, zout_lo, zout_hi); break;
    case 2: launch_tb<2>(st, g, p, np, div, zin_lo, zin_hi, zout_lo, zout_hi); break;
    case 3: launch_tb<3>(st, g, p, np, div, zin_lo, zin_hi, zout_lo, zout_hi); break;
    default: launch_tb<4>(st, g, p, np, div, zin_lo, zin_hi, zout_lo, zout_hi); break;
    }
    return true;
}

int jacobi_launches(const FsGrid& g, int sweeps, int naive)
{
    if (naive || !can_block(g)) return sweeps;
    return sweeps / kT + sweeps % kT;
}

// Single-slab driver: both z faces of the slab must be global walls.
void jacobi_multi(hipStream_t st, const FsGrid& g, float* p, float* np, const float* div,
                  int sweeps, int naive, int* result_in_np, int* prev_held)
{
    float *a = p, *b = np;
    int left = sweeps;
    *prev_held = sweeps > 0 && (naive || !can_block(g) || sweeps % kT != 0);
    if (!naive && can_block(g)) {
        for (; left >= kT; left -= kT) {
            launch_tb<kT>(st, g, a, b, div, g.z0, g.z0 + g.nzl, g.z0, g.z0 + g.nzl);
            float* t = a; a = b; b = t;
        }
    }
    for (; left > 0; --left) {
        jacobi_sweep(st, g, a, b, div, g.z0, g.z0 + g.nzl);
        float* t = a; a = b; b = t;
    }
    *result_in_np = (a == np) ? 1 : 0;
}

}  // namespace fsk
