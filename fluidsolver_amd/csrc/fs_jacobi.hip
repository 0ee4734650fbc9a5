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
// Algorithmic traffic stays 12 B per cell per sweep (read p, read div, write p').  A blocked pass
// of T = 4 sweeps moves about 18 B per cell through the L2s (measured at 512^3: 1.9 GB of reads,
// of which the tile halos and the partly used 128-byte lines at the tile edges are ~0.85 GB, and
// 0.54 GB of writes) and runs at the ~5 TB/s this chip sustains for a 2:1 read/write mix.
#include "fs_kernels.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>
#include <utility>

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
// AX = 1 variant: two halo lanes per side, tile origin shifted by 8 cells, so a tile's output rows
// are 48 cells = 192 bytes starting on a 64-byte boundary (no partly written 64-byte blocks)
constexpr int tile_bx(int ax) { return ax ? kW - 16 : kBX; }

struct TbArgs {
    const float* p;      // pre-offset: global z indexes directly
    float* np;
    const float* div;
    int nx, ny, nz;
    int pitch, plane;
    int zin_lo, zin_hi;  // level-0 planes that may be read: [zin_lo, zin_hi)
    int zout_lo, zout_hi;// planes to produce
    int zout2_lo, zout2_hi;  // optional second range of planes to produce (empty if lo >= hi)
    int chunks1;         // z chunks of the first range; chunks beyond it belong to the second
    int lz;              // output planes per workgroup along z
    int tiles_x, tiles_y, nblocks;
    int order;           // 0: plain block order, 1: XCD-contiguous (z-chunk, y, x) order
    int kpar;            // kRbgs: parity of the global z index of plane 0 (a slab's window may start on an odd plane)
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

typedef unsigned u4 __attribute__((ext_vector_type(4)));

// Buffer descriptor of plane k of a field: the march addresses memory as (scalar plane base) +
// (per-lane byte offset that never changes), so a step needs no vector address arithmetic.
// Offsets at or beyond the plane size are out of range: such a store is dropped by the hardware.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t plane_rsrc(const float* base, int k, const TbArgs& A)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + (long)k * A.plane), 0,
                                             (unsigned)A.plane * 4u, 0x00020000);
}
__device__ __forceinline__ f4 plane_load(__amdgpu_buffer_rsrc_t rs, int off)
{
    return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
}

constexpr int wrap(int a, int m) { return ((a % m) + m) % m; }

// Per-thread registers of the march.  Every array is indexed by plane number (counted from the
// workgroup's first step) modulo its ring size, and the march step is compiled once per residue
// modulo 6, so all indices are compile-time constants and no value ever moves between registers:
//   q[k % 2]        p of plane k while its load is in flight (planes n+1, n+2 at step n)
//   w0[k % 3]       p (level 0) of plane k once landed; planes n-2..n
//   w[t-1][k % 3]   level t (1..T-1) of plane k; planes n-t-2..n-t
//   dv[k % 6]       div of plane k; planes n-T..n-1 in use, n landed, n+1 in flight
template <int T, int CPT>
struct TbRegs {
    f4 q[2][CPT];
    f4 w0[3][CPT];
    f4 w[T > 1 ? T - 1 : 1][3][CPT];
    f4 dv[6][CPT];
};

// plane n+OFF of level LV at a step whose residue is PH
template <int LV, int PH, int OFF, int T, int CPT>
__device__ __forceinline__ f4& tb_plane(TbRegs<T, CPT>& R, int c)
{
    if constexpr (LV == 0) return R.w0[wrap(PH + OFF, 3)][c];
    else return R.w[LV - 1][wrap(PH + OFF, 3)][c];
}

// Per-thread constants of the march.
template <int CPT>
struct TbLane {
    int ld_off[CPT];     // byte offset in a plane, clamped into the domain: loads are always legal
    int st_off[CPT];     // byte offset in a plane; beyond the plane (store dropped) if not an output
    int row[CPT];        // own LDS row
    int row_dn[CPT];     // LDS row of the y-1 neighbour (own row at the y = 0 wall)
    int row_up[CPT];     // LDS row of the y+1 neighbour (own row at the y = ny-1 wall)
    bool at_left, at_right;
    int tx;
};

struct TbRange {
    int n_begin, n_end;  // level-0 planes this workgroup reads: [n_begin, n_end)
    int out_lo, out_hi;  // planes it stores
    int y0;              // first row of the tile
};

// What one level computes from the previous one:
//   kJacobi  a Jacobi sweep of the clamped (Neumann) pressure operator (Solver.cu:426-443)
//   kRbgs    one colour of a red/black Gauss-Seidel sweep of the zero-outside operator, the
//            multigrid smoother (Smooth, Solver.cu:848-860): level t updates the cells with
//            (x + y + z + kpar + t - 1) even to (rhs + neibor_sum) / 6 and carries the others over, so a
//            pass of T levels is T/2 sweeps, red colour first.  `div` holds the right-hand side.
//   kRbgsZero  kRbgs whose input is known to be all zero (the first smoothing of a V-cycle level):
//              level 0 is not loaded
enum { kJacobi = 0, kRbgs = 1, kRbgsZero = 2 };

// Level t of march step n: produces plane n-t of level t from planes n-t-1, n-t, n-t+1 of
// level t-1 (registers), the y-neighbours of plane n-t (LDS, written by the previous step) and div.
// Levels run on every step, also on planes outside the range the workgroup needs (before the
// ramp-up has reached them, or beyond a z wall): those results are never consumed by a plane that
// is stored, they keep the register and LDS traffic on one fixed schedule, and the step body is
// free of branches apart from the uniform wall fix-ups.
// NT: the stores of the last level carry the non-temporal cache policy (gfx950 cache-policy bit 1).
// Measured at 512^3, where a field is larger than the Infinity Cache: 7.86 -> 7.56 ms per 80 sweeps;
// at 256^3, where the three fields stay cache resident, it costs 12 %; on the loads it costs 10 % at
// 512^3.  So it is one more candidate of the timing at creation, never a rule.
template <int KIND, int T, int NTY, int CPT, int PH, int t, int NT>
__device__ __forceinline__ void tb_level(int n, TbRegs<T, CPT>& R, const TbLane<CPT>& L, const TbArgs& A,
                                         const TbRange& Z, f4* lcur, const f4* lprv)
{
    constexpr int H = NTY * CPT;
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    const int k = n - t;                                   // plane this level produces now
    if (k == A.nz - 1) {                                   // z+1 neighbour at the far wall: clamped / zero
#pragma unroll
        for (int c = 0; c < CPT; ++c)
            tb_plane<t - 1, PH, -t + 1>(R, c) = KIND == kJacobi ? tb_plane<t - 1, PH, -t>(R, c) : zero;
    }
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        const f4 ctr = tb_plane<t - 1, PH, -t>(R, c);      // previous level, plane k
        f4 pbo = lprv[((t - 1) * H + L.row_dn[c]) * kW4];  // y - 1
        f4 pt = lprv[((t - 1) * H + L.row_up[c]) * kW4];   // y + 1
        const f4 pbh = tb_plane<t - 1, PH, -t - 1>(R, c);  // z - 1
        const f4 pf = tb_plane<t - 1, PH, -t + 1>(R, c);   // z + 1
        float lft = row_prev(ctr.w), rgt = row_next(ctr.x);
        const f4 d = R.dv[wrap(PH - t, 6)][c];
        f4 r;
        if constexpr (KIND == kJacobi) {
            if (L.at_left) lft = ctr.x;
            if (L.at_right) rgt = ctr.w;
            const float s0 = lft + ctr.y + pbh.x + pf.x + pbo.x + pt.x - d.x;
            const float s1 = ctr.x + ctr.z + pbh.y + pf.y + pbo.y + pt.y - d.y;
            const float s2 = ctr.y + ctr.w + pbh.z + pf.z + pbo.z + pt.z - d.z;
            const float s3 = ctr.z + rgt + pbh.w + pf.w + pbo.w + pt.w - d.w;
            r = f4{fsk::div6(s0), fsk::div6(s1), fsk::div6(s2), fsk::div6(s3)};
        } else {
            // zero outside the box (cg_sample, Solver.cu:37-42).  A row that is its own y-neighbour
            // is a wall row or a tile-edge row; the latter is halo whose value is never used.
            if (L.at_left) lft = 0.f;
            if (L.at_right) rgt = 0.f;
            if (L.row_dn[c] == L.row[c]) pbo = zero;
            if (L.row_up[c] == L.row[c]) pt = zero;
            // rhs + neibor_sum (Solver.cu:44-50: x-1, x+1, y-1, y+1, z-1, z+1, summed left to right)
            const float s0 = d.x + (lft + ctr.y + pbo.x + pt.x + pbh.x + pf.x);
            const float s1 = d.y + (ctr.x + ctr.z + pbo.y + pt.y + pbh.y + pf.y);
            const float s2 = d.z + (ctr.y + ctr.w + pbo.z + pt.z + pbh.z + pf.z);
            const float s3 = d.w + (ctr.z + rgt + pbo.w + pt.w + pbh.w + pf.w);
            // x is a multiple of 4: the colour of this level is cells 0, 2 or cells 1, 3 of the quad
            const bool even = ((Z.y0 + L.row[c] + k + A.kpar + (t - 1)) & 1) == 0;
            const float da = fsk::div6(even ? s0 : s1), db = fsk::div6(even ? s2 : s3);
            r = even ? f4{da, ctr.y, db, ctr.w} : f4{ctr.x, da, ctr.z, db};
        }
        if constexpr (t < T) {
            tb_plane<t, PH, -t>(R, c) = r;
            lcur[(t * H + L.row[c]) * kW4] = r;
            // plane 0 of level t: its z-1 neighbour (clamped / zero), for level t+1 at the next step
            if (k == 0) tb_plane<t, PH, -t - 1>(R, c) = KIND == kJacobi ? r : zero;
        } else if (k >= Z.out_lo && k < Z.out_hi) {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, r), plane_rsrc(A.np, k, A), L.st_off[c], 0, NT ? 2 : 0);
        }
    }
}

// One march step; PH == (steps since the first one of this workgroup) mod 6.
template <int KIND, int T, int NTY, int CPT, int PH, int NT, int... Ts>
__device__ __forceinline__ void tb_step(int n, TbRegs<T, CPT>& R, const TbLane<CPT>& L, const TbArgs& A,
                                        const TbRange& Z, f4* lds, std::integer_sequence<int, Ts...>)
{
    constexpr int H = NTY * CPT;
    constexpr int cur = PH & 1, prv = cur ^ 1;             // LDS buffer written / read by this step
    f4* lcur = lds + cur * (T * H * kW4) + L.tx;
    const f4* lprv = lds + prv * (T * H * kW4) + L.tx;

    // loads run two planes (p) and one plane (div) ahead of their first use; plane numbers are
    // clamped into the readable range, so every load is legal (and a clamped p plane is exactly
    // the reference's clamped neighbour at a z wall)
    const int kp = min(max(n + 2, Z.n_begin), Z.n_end - 1);
    const int kd = min(max(n + 1, Z.n_begin), Z.n_end - 1);
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        const f4 landed = R.q[PH & 1][c];                  // plane n
        if constexpr (KIND == kRbgsZero) R.q[PH & 1][c] = f4{0.f, 0.f, 0.f, 0.f};
        else R.q[PH & 1][c] = plane_load(plane_rsrc(A.p, kp, A), L.ld_off[c]);
        R.dv[wrap(PH + 1, 6)][c] = plane_load(plane_rsrc(A.div, kd, A), L.ld_off[c]);
        R.w0[PH % 3][c] = landed;
        lcur[L.row[c] * kW4] = landed;
        // plane 0: the clamped z-1 neighbour is the plane itself
        if (n == 0) R.w0[wrap(PH - 1, 3)][c] = KIND == kJacobi ? landed : f4{0.f, 0.f, 0.f, 0.f};
    }
    (tb_level<KIND, T, NTY, CPT, PH, Ts + 1, NT>(n, R, L, A, Z, lcur, lprv), ...);
    __syncthreads();
}

template <int KIND, int T, int NTY, int CPT, int AX = 0, int NT = 0>
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
    // Tile (bx, by) covers cells [bx*kBX, bx*kBX + 64) x [by*BY, by*BY + H).  Its outer kHaloX
    // columns and T rows are halo, except where the tile edge is a wall: a wall cell needs nothing
    // from beyond the wall, so it is an output of the tile like an interior cell.
    const int x = bx * tile_bx(AX) + 4 * L.tx - (AX ? 8 : 0);    // first cell of the column
    const bool x_in = x >= 0 && x < A.nx;                        // nx % 4 == 0: all four or none
    L.at_left = x == 0;
    L.at_right = x + 4 == A.nx;
    const bool x_out = AX ? (L.tx >= 2 && L.tx < kW4 - 2)
                          : (L.tx >= 1 || L.at_left) && (L.tx < kW4 - 1 || L.at_right);
    const int xc = max(0, min(x, A.nx - 4));
    const int y0 = by * BY;
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        const int row = ty + c * NTY;
        const int y = y0 + row;
        const bool in = x_in && y < A.ny;
        L.row[c] = row;
        // clamped y-neighbours (Solver.cu:435-436): a wall row reads itself
        L.row_dn[c] = (y == 0 || row == 0) ? row : row - 1;
        L.row_up[c] = (y == A.ny - 1 || row == H - 1) ? row : row + 1;
        const bool store = in && x_out && (row >= T || y0 == 0) && (row < H - T || y0 + H >= A.ny);
        L.st_off[c] = store ? (x + y * A.pitch) * 4 : (int)0xfffffff0u;
        L.ld_off[c] = (xc + min(y, A.ny - 1) * A.pitch) * 4;
    }

    TbRange Z;
    Z.y0 = y0;
    const bool second = bz >= A.chunks1;
    Z.out_lo = second ? A.zout2_lo + (bz - A.chunks1) * A.lz : A.zout_lo + bz * A.lz;
    Z.out_hi = min(Z.out_lo + A.lz, second ? A.zout2_hi : A.zout_hi);
    Z.n_begin = max(Z.out_lo - T, A.zin_lo);
    Z.n_end = min(Z.out_hi + T, A.zin_hi);

    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    TbRegs<T, CPT> R;
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
#pragma unroll
        for (int i = 0; i < 6; ++i) R.dv[i][c] = zero;
        R.q[0][c] = R.q[1][c] = R.w0[0][c] = R.w0[1][c] = R.w0[2][c] = zero;
#pragma unroll
        for (int t = 0; t < (T > 1 ? T - 1 : 1); ++t)
#pragma unroll
            for (int i = 0; i < 3; ++i) R.w[t][i][c] = zero;
    }

    // The march runs from two steps before the first plane (those steps only issue the first
    // loads) until the last level has drained.  Ring slots are relative to the first step, so
    // the chain of six step bodies is always entered at its head.
    const int last = Z.n_end + T;
    constexpr auto levels = std::make_integer_sequence<int, T>{};
    for (int n = Z.n_begin - 2;;) {
        tb_step<KIND, T, NTY, CPT, 0, NT>(n, R, L, A, Z, lds, levels);
        if (++n >= last) break;
        tb_step<KIND, T, NTY, CPT, 1, NT>(n, R, L, A, Z, lds, levels);
        if (++n >= last) break;
        tb_step<KIND, T, NTY, CPT, 2, NT>(n, R, L, A, Z, lds, levels);
        if (++n >= last) break;
        tb_step<KIND, T, NTY, CPT, 3, NT>(n, R, L, A, Z, lds, levels);
        if (++n >= last) break;
        tb_step<KIND, T, NTY, CPT, 4, NT>(n, R, L, A, Z, lds, levels);
        if (++n >= last) break;
        tb_step<KIND, T, NTY, CPT, 5, NT>(n, R, L, A, Z, lds, levels);
        if (++n >= last) break;
    }
}


// ---- deeper blocking: six sweeps per pass ---------------------------------------------------------
//
// k_jacobi_tb keeps three planes of every intermediate level in registers (z-1, z, z+1 of the cell's
// own column), which is what limits it to four sweeps per pass.  k_jacobi_deep keeps two values per
// level instead: the level's latest plane c, and a = (x- + x+) + z-, the head of the next level's sum
// for that plane, formed when the plane is produced.  When the following plane arrives (it is the z+
// neighbour), the next level finishes the sum in the reference's order,
//     ((((pl + pr) + pbh) + pf) + pbo) + pt - div            (Solver.cu:441)
// so every value is still the reference expression of the previous level, bit for bit.  At T = 6
// that is 2 T + (T + 2) + 2 = 22 float4 of state per thread; the steady-state step body needs 152
// registers with its temporaries, which fits 12 waves per CU (168 registers), not 16 (128): tiles
// are 64 cells x 48 rows, one row of 16 lanes per tile row.  The y-neighbours go through LDS as in
// k_jacobi_tb (one plane per level, double buffered, one barrier per step: 2 x 6 x 12 KB).  In x the
// geometry is that of the 64-byte aligned variant of k_jacobi_tb (two halo quads per side: enough
// for up to eight sweeps, output rows of 48 cells starting on a 64-byte boundary), in y a tile
// loses T rows per side.  Nothing may spill: a reload from scratch is a vector-memory operation and
// waiting for it (they complete in order) would also wait for the loads of the planes ahead.
// Measured at 512^3: 488 us per pass of six sweeps against 376 us per pass of four (81 vs 94 us per
// sweep); at 256^3, where the fields stay in the Infinity Cache and the passes are bound by on-chip
// work, the four-sweep pass wins, so the choice is one more outcome of the timing at creation.
template <int T>
struct DeepRegs {
    f4 q[2];        // p of plane k while its load is in flight (planes n+1, n+2 at step n)
    f4 c[T];        // level l (0..T-1): its latest plane, n - l at the end of step n
    f4 a[T];        // (x- + x+) + z- of that plane
    f4 dv[T + 2];   // div of plane k at k % (T+2): planes n-T..n-1 in use, n landed, n+1 in flight
};

struct DeepLane {
    int ld_off, st_off;          // byte offsets in a plane (see TbLane)
    int row, row_dn, row_up;     // own LDS row and those of the clamped y-neighbours
    bool at_left, at_right;
    int tx;
};

// Level l+1 of march step n: finishes plane k = n-l-1 of level l+1 from the head a[l], the plane nw
// of level l that has just been produced (plane k+1, the z+ neighbour), the y-neighbours pbo, pt of
// plane k (read from LDS by the caller) and div; then replaces the state of level l by that of plane
// nw and hands nw to the other waves through LDS.  Returns the finished plane.
// WALLS: the step is one of the few in which some level touches a z wall (clamped neighbours,
// Solver.cu:437-438): beyond the far wall the z+ neighbour is the plane itself, and plane 0 is its
// own z- neighbour.  The steady-state body is compiled without those selects.
template <int T, int NTY, int PH, bool WALLS, int l>
__device__ __forceinline__ f4 deep_level(int n, const f4 nw, const f4 pbo, const f4 pt, DeepRegs<T>& R,
                                         const DeepLane& L, const TbArgs& A, f4* lwr)
{
    constexpr int H = NTY, RING = T + 2;
    const int k = n - l - 1;
    const f4 d = R.dv[wrap(PH - l - 1, RING)];
    const f4 h = R.a[l];
    f4 pf = nw, zm = R.c[l];
    if constexpr (WALLS) {
        const bool far = k == A.nz - 1, near = k + 1 == 0;
        pf = f4{far ? zm.x : nw.x, far ? zm.y : nw.y, far ? zm.z : nw.z, far ? zm.w : nw.w};
        zm = f4{near ? nw.x : zm.x, near ? nw.y : zm.y, near ? nw.z : zm.z, near ? nw.w : zm.w};
    }
    f4 r;
    r.x = fsk::div6((((h.x + pf.x) + pbo.x) + pt.x) - d.x);
    r.y = fsk::div6((((h.y + pf.y) + pbo.y) + pt.y) - d.y);
    r.z = fsk::div6((((h.z + pf.z) + pbo.z) + pt.z) - d.z);
    r.w = fsk::div6((((h.w + pf.w) + pbo.w) + pt.w) - d.w);
    // plane nw becomes the level's latest: the head of its sum, (pl + pr) + pbh
    float lft = row_prev(nw.w), rgt = row_next(nw.x);
    if (L.at_left) lft = nw.x;
    if (L.at_right) rgt = nw.w;
    float ax = (lft + nw.y) + zm.x, ay = (nw.x + nw.z) + zm.y, az = (nw.y + nw.w) + zm.z, aw = (nw.z + rgt) + zm.w;
    // The head is not used before the next step.  Pin its evaluation here: left to itself the
    // compiler sinks the heads of all levels to the end of the step, which keeps the old and the new
    // plane of every level alive at once and spills.
    asm volatile("" : "+v"(ax), "+v"(ay), "+v"(az), "+v"(aw));
    R.a[l] = f4{ax, ay, az, aw};
    R.c[l] = nw;
    lwr[l * H * kW4] = nw;       // into the buffer the next step reads
    return r;
}

// levels l .. T-1 of a step; the LDS reads of a level are issued before the level below it computes
template <int T, int NTY, int PH, bool WALLS, int l>
__device__ __forceinline__ f4 deep_levels(int n, const f4 nw, const f4 pbo, const f4 pt, DeepRegs<T>& R,
                                          const DeepLane& L, const TbArgs& A, const f4* lrd, f4* lwr)
{
    constexpr int H = NTY;
    if constexpr (l + 1 < T) {
        const f4 pbo2 = lrd[((l + 1) * H + L.row_dn) * kW4];
        const f4 pt2 = lrd[((l + 1) * H + L.row_up) * kW4];
        const f4 r = deep_level<T, NTY, PH, WALLS, l>(n, nw, pbo, pt, R, L, A, lwr);
        return deep_levels<T, NTY, PH, WALLS, l + 1>(n, r, pbo2, pt2, R, L, A, lrd, lwr);
    } else {
        return deep_level<T, NTY, PH, WALLS, l>(n, nw, pbo, pt, R, L, A, lwr);
    }
}

template <int T, int NTY, int PH, int NT, bool WALLS>
__device__ __forceinline__ void deep_step(int n, DeepRegs<T>& R, const DeepLane& L, const TbArgs& A, const TbRange& Z,
                                          f4* lds)
{
    constexpr int H = NTY, RING = T + 2;
    static_assert(RING % 2 == 0, "the load slots and the LDS buffers alternate with the ring");
    constexpr int cur = PH & 1, prv = cur ^ 1;              // LDS buffer written / read by this step
    const f4* lrd = lds + prv * (T * H * kW4) + L.tx;
    f4* lwr = lds + cur * (T * H * kW4) + L.row * kW4 + L.tx;
    const f4 pbo = lrd[L.row_dn * kW4], pt = lrd[L.row_up * kW4];
    const int kp = min(max(n + 2, Z.n_begin), Z.n_end - 1);
    const int kd = min(max(n + 1, Z.n_begin), Z.n_end - 1);
    f4 nw = R.q[PH & 1];                                    // plane n of level 0 has landed
    R.q[PH & 1] = plane_load(plane_rsrc(A.p, kp, A), L.ld_off);
    R.dv[wrap(PH + 1, RING)] = plane_load(plane_rsrc(A.div, kd, A), L.ld_off);
    nw = deep_levels<T, NTY, PH, WALLS, 0>(n, nw, pbo, pt, R, L, A, lrd, lwr);
    const int k = n - T;                                    // plane of level T that is complete now
    if (k >= Z.out_lo && k < Z.out_hi)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, nw), plane_rsrc(A.np, k, A), L.st_off, 0, NT ? 2 : 0);
    __syncthreads();
}

// The march chains the T + 2 = 8 step bodies (one per residue of the ring indices).
// deep_run: steps n .. end-1 entering the chain at its head (n and the chain in phase); returns the
// body the step after the last one takes.  deep_finish: steps n .. end-1 entering the chain at body ph.
// Both leave n = end.  The chain entered at its head is what the steady state runs: its bodies are
// reached in one way only, so the register assignment rotates through them without copies; the
// chain with entry points at every body is for the few closing steps.
template <int T, int NTY, int NT, bool WALLS>
__device__ __forceinline__ int deep_run(int& n, int end, DeepRegs<T>& R, const DeepLane& L, const TbArgs& A,
                                        const TbRange& Z, f4* lds)
{
    static_assert(T + 2 == 8, "eight step bodies");
    if (n >= end) return 0;
#define FS_DEEP_STEP(PH)                                           \
    deep_step<T, NTY, PH, NT, WALLS>(n, R, L, A, Z, lds);  \
    if (++n >= end) return (PH + 1) & 7;
    for (;;) {
        FS_DEEP_STEP(0)
        FS_DEEP_STEP(1)
        FS_DEEP_STEP(2)
        FS_DEEP_STEP(3)
        FS_DEEP_STEP(4)
        FS_DEEP_STEP(5)
        FS_DEEP_STEP(6)
        FS_DEEP_STEP(7)
    }
#undef FS_DEEP_STEP
}

template <int T, int NTY, int NT, bool WALLS>
__device__ __forceinline__ void deep_finish(int& n, int ph, int end, DeepRegs<T>& R, const DeepLane& L, const TbArgs& A,
                                            const TbRange& Z, f4* lds)
{
    if (n >= end) return;
#define FS_DEEP_STEP(PH)                                               \
    case PH:                                                           \
        deep_step<T, NTY, PH, NT, WALLS>(n, R, L, A, Z, lds);  \
        if (++n >= end) return;                                        \
        [[fallthrough]];
    for (;;) {
        switch (ph) {
            FS_DEEP_STEP(0)
            FS_DEEP_STEP(1)
            FS_DEEP_STEP(2)
            FS_DEEP_STEP(3)
            FS_DEEP_STEP(4)
            FS_DEEP_STEP(5)
            FS_DEEP_STEP(6)
            FS_DEEP_STEP(7)
        default: break;
        }
        ph = 0;
    }
#undef FS_DEEP_STEP
}

template <int T, int NTY, int NT>
__global__ void __launch_bounds__(kW4 * NTY)
k_jacobi_deep(TbArgs A)
{
    constexpr int H = NTY, BY = H - 2 * T, RING = T + 2;
    static_assert(T > kHaloX && T <= 2 * kHaloX, "two halo quads per side in x");
    static_assert(BY > 0, "tile too small");
    static_assert(RING == 8, "the march below chains eight step bodies");
    __shared__ f4 lds[2 * T * H * kW4];

    DeepLane L;
    L.tx = threadIdx.x;
    const int row = threadIdx.y;
    const int lid = A.order ? xcd_contiguous((int)blockIdx.x, A.nblocks) : (int)blockIdx.x;
    const int bz = lid / (A.tiles_x * A.tiles_y);
    const int rem = lid - bz * (A.tiles_x * A.tiles_y);
    const int by = rem / A.tiles_x;
    const int bx = rem - by * A.tiles_x;
    // tile (bx, by) covers cells [bx*48 - 8, bx*48 + 56) x [by*BY, by*BY + H), H = NTY rows; the
    // outer two quads and T rows are halo except where the tile edge is a wall
    const int x = bx * tile_bx(1) + 4 * L.tx - 8;
    const bool x_in = x >= 0 && x < A.nx;
    L.at_left = x == 0;
    L.at_right = x + 4 == A.nx;
    const bool x_out = L.tx >= 2 && L.tx < kW4 - 2;
    const int xc = max(0, min(x, A.nx - 4));
    const int y0 = by * BY, y = y0 + row;
    L.row = row;
    L.row_dn = (y == 0 || row == 0) ? row : row - 1;
    L.row_up = (y == A.ny - 1 || row == H - 1) ? row : row + 1;
    const bool store = x_in && y < A.ny && x_out && (row >= T || y0 == 0) && (row < H - T || y0 + H >= A.ny);
    L.st_off = store ? (x + y * A.pitch) * 4 : (int)0xfffffff0u;
    L.ld_off = (xc + min(y, A.ny - 1) * A.pitch) * 4;

    TbRange Z;
    Z.y0 = y0;
    const bool second = bz >= A.chunks1;
    Z.out_lo = second ? A.zout2_lo + (bz - A.chunks1) * A.lz : A.zout_lo + bz * A.lz;
    Z.out_hi = min(Z.out_lo + A.lz, second ? A.zout2_hi : A.zout_hi);
    Z.n_begin = max(Z.out_lo - T, A.zin_lo);
    Z.n_end = min(Z.out_hi + T, A.zin_hi);

    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    DeepRegs<T> R;
    R.q[0] = R.q[1] = zero;
#pragma unroll
    for (int i = 0; i < T; ++i) R.c[i] = R.a[i] = zero;
#pragma unroll
    for (int i = 0; i < RING; ++i) R.dv[i] = zero;
    // the LDS planes are read (by the levels, which run on every step) before the first step has
    // written them: start from zeros, so that no step computes on uninitialised memory
#pragma unroll
    for (int i = 0; i < 2 * T; ++i) lds[(i * H + row) * kW4 + L.tx] = zero;
    __syncthreads();

    // The march runs from two steps before the first plane (those steps only issue the first loads)
    // until the last level has drained.  A level touches a z wall only in the first T steps of a
    // march that starts at the near wall and in the last T steps of one that reaches the far wall:
    // those steps take the step bodies with the wall selects, all others the plain ones.
    const int last = Z.n_end + T;
    int n = Z.n_begin - 2;
    // (a march from the near wall: n = -2 .. T-1, exactly once through the chain, T + 2 <= last)
    if (Z.n_begin == 0) deep_run<T, NTY, NT, true>(n, T, R, L, A, Z, lds);
    const int ph = deep_run<T, NTY, NT, false>(n, Z.n_end == A.nz ? min(A.nz, last) : last, R, L, A, Z, lds);
    deep_finish<T, NTY, NT, true>(n, ph, last, R, L, A, Z, lds);
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
constexpr TbConfig kConfigs[] = {{64, 1}, {32, 1}, {64, 1}, {32, 1}};   // 2, 3: 64-byte aligned output rows
constexpr int kNumConfigs = sizeof(kConfigs) / sizeof(kConfigs[0]);

// Tile variant for a grid: 64-row tiles (one workgroup of 16 waves per CU) have the smaller halo
// share and win on large planes; 32-row tiles (two workgroups per CU) give more workgroups per
// plane and win on small ones.  Measured per 80 sweeps, 32-row vs 64-row tiles: 128^3 0.38 vs
// 0.56 ms, 192^3 0.667 vs 0.670, 256^3 1.11 vs 1.05, 384^3 4.16 vs 3.63.  FS_JACOBI_CFG overrides.
int tb_config(const FsGrid& g)
{
    if (g.env.jacobi_cfg >= 0 && g.env.jacobi_cfg < kNumConfigs) return g.env.jacobi_cfg;
    if (g.jacobi_tile > 0 && g.jacobi_tile <= kNumConfigs) return g.jacobi_tile - 1;
    return (long)g.nx * g.ny >= 224L * 224L ? 0 : 1;
}

// FS_JACOBI_NT = 0 / 1 forces the store policy
int tb_nt(const FsGrid& g) { return g.env.jacobi_nt >= 0 ? (g.env.jacobi_nt != 0) : g.jacobi_nt; }

template <int KIND, int T, int NTY, int CPT, int AX = 0>
void launch_tb_cfg(hipStream_t st, const FsGrid& g, TbArgs A)
{
    constexpr int H = NTY * CPT, BY = H - 2 * T;
    // halo columns / rows at a wall are outputs, so the tiles need to cover n - 2*halo only
    const int tiles_x = AX ? cdiv(g.nx, tile_bx(1)) : max(1, cdiv(g.nx - 2 * kHaloX, kBX));
    const int tiles_y = max(1, cdiv(g.ny - 2 * T, BY));
    int tiles = tiles_x * tiles_y;
    const int n1 = A.zout_hi - A.zout_lo, n2 = max(0, A.zout2_hi - A.zout2_lo);
    int nout = n1 + n2;
    // Cut z into chunks.  A workgroup costs (planes + 2T + 2) march steps (ramp-up and drain at
    // each chunk edge).  With `slots` workgroups resident at once, a launch of up to about two
    // rounds takes ceil(blocks / slots) rounds; with more rounds the workgroups drift apart, the
    // tail round overlaps the others and the memory system is shared more evenly (measured ~5 %
    // faster per march step at 512^3), so the cost is the fractional round count.
    constexpr int lds_bytes = 2 * T * H * kW4 * 16;
    constexpr int per_cu_lds = 163840 / lds_bytes;
    constexpr int per_cu_waves = 16 / (kW4 * NTY / 64 > 16 ? 16 : kW4 * NTY / 64);   // <= 4 waves/SIMD at ~128 VGPRs
    const int slots = 256 * max(1, min(per_cu_lds, per_cu_waves));
    int chunks = 1;
    double best = -1.0;
    for (int c = 1; c <= max(1, nout / 8); ++c) {
        const double r = (double)tiles * c / slots;
        const double steps = cdiv(nout, c) + 2 * T + 2;
        double cost;
        if (r <= 1.0) cost = steps / r;              // idle CUs in the only round are paid for too
        else if (r < 2.5) cost = cdiv(tiles * c, slots) * steps;
        else cost = 0.95 * r * steps;
        if (best < 0 || cost < best) {
            best = cost;
            chunks = c;
        }
    }
    A.lz = cdiv(nout, chunks);
    if (g.jacobi_lz > 0) A.lz = min(nout, g.jacobi_lz);
    if (g.env.jacobi_lz > 0) A.lz = min(nout, g.env.jacobi_lz);
    A.chunks1 = cdiv(n1, A.lz);      // chunks never straddle the two ranges
    chunks = A.chunks1 + cdiv(n2, A.lz);
    A.tiles_x = tiles_x;
    A.tiles_y = tiles_y;
    A.nblocks = A.tiles_x * A.tiles_y * chunks;
    A.order = g.env.jacobi_order;
    dim3 grid(A.nblocks), block(kW4, NTY);
    if constexpr (KIND == kJacobi && T == kT) {
        if (tb_nt(g)) {
            hipLaunchKernelGGL((k_jacobi_tb<KIND, T, NTY, CPT, AX, 1>), grid, block, 0, st, A);
            return;
        }
    }
    hipLaunchKernelGGL((k_jacobi_tb<KIND, T, NTY, CPT, AX>), grid, block, 0, st, A);
}

template <int T, int KIND = kJacobi>
void launch_tb(hipStream_t st, const FsGrid& g, const float* p, float* np, const float* div,
               int zin_lo, int zin_hi, int zout_lo, int zout_hi, int zout2_lo = 0, int zout2_hi = 0, int kpar = 0)
{
    TbArgs A;
    A.kpar = kpar & 1;
    ptrdiff_t zoff = (ptrdiff_t)g.z0 * g.plane;
    A.p = p - zoff;
    A.np = np - zoff;
    A.div = div - zoff;
    A.nx = g.nx; A.ny = g.ny; A.nz = g.nz;
    A.pitch = g.pitch; A.plane = g.plane;
    A.zin_lo = zin_lo; A.zin_hi = zin_hi;
    A.zout_lo = zout_lo; A.zout_hi = zout_hi;
    A.zout2_lo = zout2_lo; A.zout2_hi = zout2_hi;
    A.chunks1 = 0;
    A.lz = 0;
    switch (tb_config(g)) {
    case 1: launch_tb_cfg<KIND, T, 32, 1>(st, g, A); break;
    case 2: launch_tb_cfg<KIND, T, 64, 1, 1>(st, g, A); break;
    case 3: launch_tb_cfg<KIND, T, 32, 1, 1>(st, g, A); break;
    default: launch_tb_cfg<KIND, T, 64, 1>(st, g, A); break;
    }
}

constexpr int kTDeep = 6;      // sweeps fused per pass of k_jacobi_deep
constexpr int kDeepRows = 48;  // its tile rows = rows of threads of a workgroup (12 waves)

bool can_deep(const FsGrid& g) { return can_block(g) && g.ny >= 2 * kTDeep + 2 && g.nzl >= 2 * kTDeep; }

// A z-slab runs six-sweep passes if every rank can (ghost planes six deep, slabs thick enough for two
// boundary layers of up to 2 kT planes and an interior) and the fields of a rank do not stay in the
// 256 MiB Infinity Cache -- where they do, the passes are bound by on-chip work and four sweeps per
// pass win.  Everything this reads is the same on every rank.
bool deep_slab(const FsGrid& g)
{
    if (g.wall_lo && g.wall_hi) return false;
    if (!can_block(g) || g.ny < 2 * kTDeep + 2 || g.pghost < kTDeep || g.nz_min < 4 * kT + 1) return false;
    if (g.env.jacobi_deep >= 0) return g.env.jacobi_deep != 0;
    return (long)g.nx * g.ny >= 224L * 224L && 3L * g.plane * g.nz_min * (long)sizeof(float) > (256L << 20);
}

// one pass of kTDeep sweeps: reads level-0 planes [zin_lo, zin_hi), writes [zout_lo, zout_hi) and the
// optional second range (see launch_tb)
void launch_deep(hipStream_t st, const FsGrid& g, const float* p, float* np, const float* div,
                 int zin_lo, int zin_hi, int zout_lo, int zout_hi, int zout2_lo = 0, int zout2_hi = 0)
{
    constexpr int T = kTDeep, NTY = kDeepRows, BY = NTY - 2 * T;
    TbArgs A;
    A.kpar = 0;
    ptrdiff_t zoff = (ptrdiff_t)g.z0 * g.plane;
    A.p = p - zoff;
    A.np = np - zoff;
    A.div = div - zoff;
    A.nx = g.nx; A.ny = g.ny; A.nz = g.nz;
    A.pitch = g.pitch; A.plane = g.plane;
    A.zin_lo = zin_lo; A.zin_hi = zin_hi;
    A.zout_lo = zout_lo; A.zout_hi = zout_hi;
    A.zout2_lo = zout2_lo; A.zout2_hi = zout2_hi;
    const int n1 = zout_hi - zout_lo, n2 = max(0, zout2_hi - zout2_lo);
    const int nout = n1 + n2;
    const int tiles_x = cdiv(g.nx, tile_bx(1)), tiles_y = max(1, cdiv(g.ny - 2 * T, BY));
    // z chunks: as for k_jacobi_tb, one workgroup per CU
    int chunks = 1;
    double best = -1.0;
    for (int c = 1; c <= max(1, nout / 12); ++c) {
        const double r = (double)tiles_x * tiles_y * c / 256.0;
        const double steps = cdiv(nout, c) + 2 * T + 2;
        const double cost = r <= 1.0 ? steps / r : (r < 2.5 ? cdiv(tiles_x * tiles_y * c, 256) * steps : 0.95 * r * steps);
        if (best < 0 || cost < best) {
            best = cost;
            chunks = c;
        }
    }
    A.lz = cdiv(nout, chunks);
    if (g.jacobi_deep_lz > 0) A.lz = min(nout, g.jacobi_deep_lz);
    if (g.env.jacobi_deep_lz > 0) A.lz = min(nout, g.env.jacobi_deep_lz);
    A.chunks1 = cdiv(n1, A.lz);      // chunks never straddle the two ranges
    chunks = A.chunks1 + cdiv(n2, A.lz);
    A.tiles_x = tiles_x;
    A.tiles_y = tiles_y;
    A.nblocks = tiles_x * tiles_y * chunks;
    A.order = g.env.jacobi_order;
    dim3 grid(A.nblocks), block(kW4, NTY);
    if (g.env.jacobi_deep_nt >= 0 ? g.env.jacobi_deep_nt != 0 : g.jacobi_deep_nt != 0)
        hipLaunchKernelGGL((k_jacobi_deep<T, NTY, 1>), grid, block, 0, st, A);
    else hipLaunchKernelGGL((k_jacobi_deep<T, NTY, 0>), grid, block, 0, st, A);
}

// the same over a single domain (both z faces global walls)
void launch_deep(hipStream_t st, const FsGrid& g, const float* p, float* np, const float* div)
{
    launch_deep(st, g, p, np, div, g.z0, g.z0 + g.nzl, g.z0, g.z0 + g.nzl);
}

// Six-sweep passes among `sweeps` sweeps when the rest runs in passes of `unit` sweeps: as many as
// leave a remainder such passes take whole, if giving up at most one achieves that; else as many as fit.
int deep_count(int sweeps, int unit)
{
    const int most = sweeps / kTDeep;
    for (int d = most; d >= 0 && d >= most - 1; --d)
        if ((sweeps - d * kTDeep) % unit == 0) return d;
    return most;
}

// How `sweeps` sweeps are cut into passes of kTDeep and kT sweeps and single sweeps when the deep
// passes are on.
void deep_split(int sweeps, int* deep, int* four, int* single)
{
    *deep = deep_count(sweeps, kT);
    const int left = sweeps - *deep * kTDeep;
    *four = left / kT;
    *single = left % kT;
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

// Which variant wins depends on how the tile counts round against the CU count and on whether the
// fields live in HBM or in the Infinity Cache (64-byte aligned output rows: 512^3 8.72 -> 8.37 ms
// per 80 sweeps, 448^3 5.08 -> 6.32), and the cost model of the z chunking does not see that ~200
// workgroups already saturate the memory system (512^3, aligned rows: 2 chunks 7.85 ms, the
// model's 7 chunks 8.35), so both are measured on the grid itself: a few passes per candidate.
// The timing loop: `pass()` launches one pass with g's current jacobi_tile / jacobi_lz over `nout`
// output planes.  Leaves the winner in g (zeros = the rule and the cost model).
template <class Pass>
void tune_tb(hipStream_t st, FsGrid& g, int nout, const char* what, Pass&& pass, bool try_nt = false)
{
    g.jacobi_tile = 0;
    g.jacobi_lz = 0;
    g.jacobi_nt = 0;
    if (g.env.jacobi_tune == 0) return;
    const bool verbose = g.env.jacobi_tune == 2;   // FS_JACOBI_TUNE=2 prints the readings
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess) return;
    if (hipEventCreate(&b) != hipSuccess) {
        (void)hipEventDestroy(a);
        return;
    }
    const bool small = (long)g.nx * g.ny < 224L * 224L;
    const int cand_small[] = {1, 3}, cand_large[] = {0, 2, 3, 1};
    const int* cand = small ? cand_small : cand_large;
    const int ncand = small ? 2 : 4;
    const int chunk_counts[] = {0, 1, 2, 3, 4, 5, 6, 8};   // 0: the cost model's choice
    // short passes need more repetitions for a stable reading
    const int timed = (long)g.nx * g.ny * nout >= 384L * 384L * 384L ? 3 : 12;
    // the rule's variant with the model's chunking is the reference: another candidate has to
    // beat it by more than the noise of these short measurements to be taken
    const int rule_cfg = small ? 1 : 0;
    float best = -1.f, rule_ms = -1.f;
    int best_cfg = -1, best_lz = 0;
    bool failed = false;
    for (int c = 0; c < ncand && !failed; ++c) {
        for (int cc : chunk_counts) {
            const int lz = cc ? cdiv(nout, cc) : 0;
            if (cc && lz < 2 * kT) continue;   // chunks shorter than their ramp never win
            g.jacobi_tile = cand[c] + 1;
            g.jacobi_lz = lz;
            pass();
            (void)hipEventRecord(a, st);
            for (int i = 0; i < timed; ++i) pass();
            (void)hipEventRecord(b, st);
            float ms = 0.f;
            if (hipEventSynchronize(b) != hipSuccess || hipEventElapsedTime(&ms, a, b) != hipSuccess) {
                failed = true;
                break;
            }
            if (verbose) fprintf(stderr, "%s %dx%dx%d: variant %d, %d planes per chunk: %.1f us per pass\n", what,
                                 g.nx, g.ny, nout, cand[c], lz, 1e3f * ms / timed);
            if (cand[c] == rule_cfg && lz == 0) rule_ms = ms;
            if (best < 0.f || ms < best) {
                best = ms;
                best_cfg = cand[c];
                best_lz = lz;
            }
        }
    }
    if (failed || best_cfg < 0 || (rule_ms > 0.f && best > 0.97f * rule_ms)) {
        g.jacobi_tile = 0;
        g.jacobi_lz = 0;
        best = rule_ms;
    } else {
        g.jacobi_tile = best_cfg + 1;
        g.jacobi_lz = best_lz;
    }
    // non-temporal stores on top of the choice so far (Jacobi passes only; pays where the fields
    // do not fit the Infinity Cache)
    if (try_nt && !failed && best > 0.f) {
        // (the chunking is searched again: the policy shifts the best one, 2 -> 5 chunks at 512^3)
        const int keep_lz = g.jacobi_lz;
        float best_nt = -1.f;
        int best_nt_lz = keep_lz;
        g.jacobi_nt = 1;
        for (int cc : chunk_counts) {
            const int lz = cc ? cdiv(nout, cc) : keep_lz;
            if (cc && (lz < 2 * kT || lz == keep_lz)) continue;
            g.jacobi_lz = lz;
            pass();
            (void)hipEventRecord(a, st);
            for (int i = 0; i < timed; ++i) pass();
            (void)hipEventRecord(b, st);
            float ms = 0.f;
            if (hipEventSynchronize(b) != hipSuccess || hipEventElapsedTime(&ms, a, b) != hipSuccess) {
                best_nt = -1.f;
                break;
            }
            if (verbose) fprintf(stderr, "%s %dx%dx%d: non-temporal stores, %d planes per chunk: %.1f us per pass "
                                 "(without: %.1f)\n", what, g.nx, g.ny, nout, lz, 1e3f * ms / timed, 1e3f * best / timed);
            if (best_nt < 0.f || ms < best_nt) {
                best_nt = ms;
                best_nt_lz = lz;
            }
        }
        if (best_nt > 0.f && best_nt <= 0.98f * best) {
            g.jacobi_lz = best_nt_lz;
        } else {
            g.jacobi_nt = 0;
            g.jacobi_lz = keep_lz;
        }
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
}

// Times the six-sweep pass against the four-sweep pass g is set up with (per sweep), over chunk
// counts and both store policies.
// [flo, fhi) / [dlo, dhi): the planes a four-sweep / six-sweep pass of this grid produces in one launch
// (the whole domain, or the interior planes of a slab).
static void tune_deep(hipStream_t st, FsGrid& g, float* x, float* y, const float* div, int flo, int fhi, int dlo, int dhi,
                      int* deep, int* deep_lz, int* deep_nt)
{
    *deep = *deep_lz = *deep_nt = 0;
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess) return;
    if (hipEventCreate(&b) != hipSuccess) {
        (void)hipEventDestroy(a);
        return;
    }
    const bool verbose = g.env.jacobi_tune == 2;
    const int nout = dhi - dlo, timed = (long)g.nx * g.ny * nout >= 384L * 384L * 384L ? 3 : 12;
    auto time_of = [&](auto&& pass) {
        pass();
        (void)hipEventRecord(a, st);
        for (int i = 0; i < timed; ++i) pass();
        (void)hipEventRecord(b, st);
        float ms = -1.f;
        if (hipEventSynchronize(b) != hipSuccess || hipEventElapsedTime(&ms, a, b) != hipSuccess) return -1.f;
        return ms / timed;
    };
    const float four = time_of([&] {
        launch_tb<kT>(st, g, x, y, div, g.z0, g.z0 + g.nzl, flo, fhi);
        float* t = x; x = y; y = t;
    });
    float best = -1.f;
    const int chunk_counts[] = {0, 1, 2, 3, 4, 5, 6, 8};
    for (int nt = 0; nt < 2; ++nt)
        for (int cc : chunk_counts) {
            const int lz = cc ? cdiv(nout, cc) : 0;
            if (cc && lz < 2 * kTDeep) continue;
            g.jacobi_deep_lz = lz;
            g.jacobi_deep_nt = nt;
            const float ms = time_of([&] {
                launch_deep(st, g, x, y, div, g.z0, g.z0 + g.nzl, dlo, dhi);
                float* t = x; x = y; y = t;
            });
            if (verbose) fprintf(stderr, "jacobi_autotune %dx%dx%d: six-sweep pass, %d planes per chunk, nt %d: %.1f us "
                                 "(four-sweep pass: %.1f)\n", g.nx, g.ny, nout, lz, nt, 1e3f * ms, 1e3f * four);
            if (ms > 0.f && (best < 0.f || ms < best)) {
                best = ms;
                *deep_lz = lz;
                *deep_nt = nt;
            }
        }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    g.jacobi_deep_lz = g.jacobi_deep_nt = 0;
    // a sweep of the six-sweep pass has to be cheaper by more than the noise of these readings (per
    // plane produced: a slab's six-sweep interior is four planes shorter than its four-sweep one)
    const float per4 = four / kT / (fhi - flo), per6 = best / kTDeep / (dhi - dlo);
    *deep = g.env.jacobi_deep == 1 || (best > 0.f && four > 0.f && per6 < 0.97f * per4);
}

void jacobi_autotune(hipStream_t st, FsGrid& g, float* p, float* np, const float* div)
{
    g.jacobi_tile = 0;
    g.jacobi_lz = 0;
    g.jacobi_nt = 0;
    g.jacobi_deep = g.jacobi_deep_lz = g.jacobi_deep_nt = 0;
    // FS_JACOBI_DEEP=1 forces the six-sweep passes wherever they can run (tests), also without timing;
    // on a slab they are on or off by a rule every rank evaluates alike, and only their chunking and
    // store policy are timed
    const bool single = g.wall_lo && g.wall_hi, slab_deep = deep_slab(g);
    const bool forced = single && g.env.jacobi_deep == 1 && can_deep(g);
    if (forced || slab_deep) g.jacobi_deep = 1;
    // what a slab's interior launch produces: planes whose T-sweep cone stays inside the owned
    // planes (a pass may only store planes it holds T input planes around, except at a wall)
    const int out_lo = g.z0 + (g.wall_lo ? 0 : kT), out_hi = g.z0 + g.nzl - (g.wall_hi ? 0 : kT);
    if (!can_block(g) || out_hi <= out_lo) return;
    // one timing per grid shape and process: solvers created later on the same shape take the
    // remembered choice, so they cost no timing passes and run the same variant
    static std::mutex mu;
    static std::map<std::tuple<int, int, int, int, int>, std::tuple<int, int, int, int, int, int>> tuned;
    std::lock_guard<std::mutex> lock(mu);
    if (g.env.jacobi_tune == 0) return;   // (and a solver created without the switch still gets its timing)
    const auto key = std::make_tuple(g.nx, g.ny, g.nzl, g.wall_lo, g.wall_hi);
    auto it = tuned.find(key);
    if (it == tuned.end()) {
        float *x = p, *y = np;
        tune_tb(st, g, out_hi - out_lo, "jacobi_autotune", [&] {
            launch_tb<kT>(st, g, x, y, div, g.z0, g.z0 + g.nzl, out_lo, out_hi);
            float* t = x; x = y; y = t;
        }, true);
        // the six-sweep passes (planes large enough for 64-row tiles): their own chunking and store
        // policy; a single domain then keeps them if a sweep is cheaper than in the best four-sweep pass
        int deep = 0, deep_lz = 0, deep_nt = 0;
        const int dlo = g.z0 + (g.wall_lo ? 0 : kTDeep), dhi = g.z0 + g.nzl - (g.wall_hi ? 0 : kTDeep);
        if (dhi - dlo >= 2 * kTDeep && (long)g.nx * g.ny >= 224L * 224L &&
            (single ? g.env.jacobi_deep != 0 && can_deep(g) : slab_deep))
            tune_deep(st, g, x, y, div, out_lo, out_hi, dlo, dhi, &deep, &deep_lz, &deep_nt);
        it = tuned.emplace(key, std::make_tuple(g.jacobi_tile, g.jacobi_lz, g.jacobi_nt, deep, deep_lz, deep_nt)).first;
    }
    g.jacobi_tile = std::get<0>(it->second);
    g.jacobi_lz = std::get<1>(it->second);
    g.jacobi_nt = std::get<2>(it->second);
    g.jacobi_deep = std::get<3>(it->second);
    g.jacobi_deep_lz = std::get<4>(it->second);
    g.jacobi_deep_nt = std::get<5>(it->second);
    // a slab: the rule decides, whatever this rank's timing said
    if (forced) g.jacobi_deep = 1;
    if (!single) g.jacobi_deep = slab_deep ? 1 : 0;
    if (!g.jacobi_deep) g.jacobi_deep_lz = g.jacobi_deep_nt = 0;
}

bool jacobi_can_block(const FsGrid& g) { return can_block(g); }

int jacobi_nontemporal(const FsGrid& g) { return g.jacobi_nt; }

void jacobi_config(const FsGrid& g, int* variant, int* rows, int* aligned, int* planes_per_chunk)
{
    const int v = can_block(g) ? tb_config(g) : -1;
    *variant = v;
    *rows = v >= 0 ? kConfigs[v].nty * kConfigs[v].cpt : 0;
    *aligned = v >= 2 ? 1 : 0;
    *planes_per_chunk = g.jacobi_lz;
}

void jacobi_deep_config(const FsGrid& g, int* sweeps_per_pass, int* planes_per_chunk, int* nontemporal)
{
    const bool on = g.jacobi_deep && ((g.wall_lo && g.wall_hi) ? can_deep(g) : deep_slab(g));
    *sweeps_per_pass = on ? kTDeep : (can_block(g) ? kT : 1);
    *planes_per_chunk = on ? g.jacobi_deep_lz : 0;
    *nontemporal = on ? g.jacobi_deep_nt : 0;
}

bool jacobi_pass(hipStream_t st, const FsGrid& g, const float* p, float* np, const float* div,
                 int sweeps, int zin_lo, int zin_hi, int zout_lo, int zout_hi, int zout2_lo, int zout2_hi)
{
    if (!can_block(g) || sweeps < 1 || sweeps > kT) return false;
    if (zout_hi <= zout_lo) {   // only the second range, or nothing
        zout_lo = zout2_lo; zout_hi = zout2_hi;
        zout2_lo = zout2_hi = 0;
    }
    if (zout_hi <= zout_lo) return true;
    switch (sweeps) {
    case 1: launch_tb<1>(st, g, p, np, div, zin_lo, zin_hi, zout_lo, zout_hi, zout2_lo, zout2_hi); break;
    case 2: launch_tb<2>(st, g, p, np, div, zin_lo, zin_hi, zout_lo, zout_hi, zout2_lo, zout2_hi); break;
    case 3: launch_tb<3>(st, g, p, np, div, zin_lo, zin_hi, zout_lo, zout_hi, zout2_lo, zout2_hi); break;
    default: launch_tb<4>(st, g, p, np, div, zin_lo, zin_hi, zout_lo, zout_hi, zout2_lo, zout2_hi); break;
    }
    return true;
}

bool jacobi_deep_pass(hipStream_t st, const FsGrid& g, const float* p, float* np, const float* div,
                      int zin_lo, int zin_hi, int zout_lo, int zout_hi, int zout2_lo, int zout2_hi)
{
    if (!can_block(g) || g.ny < 2 * kTDeep + 2) return false;
    if (zout_hi <= zout_lo) {   // only the second range, or nothing
        zout_lo = zout2_lo; zout_hi = zout2_hi;
        zout2_lo = zout2_hi = 0;
    }
    if (zout_hi <= zout_lo) return true;
    launch_deep(st, g, p, np, div, zin_lo, zin_hi, zout_lo, zout_hi, zout2_lo, zout2_hi);
    return true;
}

int jacobi_deep_block() { return kTDeep; }

bool jacobi_deep_slab(const FsGrid& g) { return g.jacobi_deep && deep_slab(g); }

int jacobi_deep_count(int sweeps, int unit) { return deep_count(sweeps, unit); }

// Red/black Gauss-Seidel on a dense nx x ny x nz array (rows unpadded): `phases` half-sweeps (2 or 4,
// red first) from zin into zout; false if this grid cannot take the blocked kernel.
bool rbgs_pass(hipStream_t st, const FsEnv& env, int nx, int ny, int nz, const float* rhs, const float* zin, float* zout,
               int phases, bool zin_is_zero, int kpar)
{
    FsGrid g{};
    g.env = env;
    g.nx = nx; g.ny = ny; g.nz = nz;
    g.z0 = 0; g.nzl = nz;
    g.pitch = nx; g.plane = nx * ny;
    g.wall_lo = g.wall_hi = 1;
    const bool aligned = (((uintptr_t)rhs | (uintptr_t)zin | (uintptr_t)zout) & 15u) == 0;
    if (!can_block(g) || !aligned || (phases != 2 && phases != 4)) return false;
    // tile variant and chunking of a level: timed the first time a level of these dimensions is
    // smoothed (the pass is repeated from the same input into the same output, which changes
    // nothing), then remembered
    {
        static std::mutex mu;
        static std::map<std::tuple<int, int, int>, std::pair<int, int>> tuned;
        std::lock_guard<std::mutex> lock(mu);
        const auto key = std::make_tuple(nx, ny, nz);
        auto it = tuned.find(key);
        if (it == tuned.end() && env.jacobi_tune != 0) {
            tune_tb(st, g, nz, "rbgs_autotune", [&] { launch_tb<4, kRbgs>(st, g, zin, zout, rhs, 0, nz, 0, nz, 0, 0, kpar); });
            it = tuned.emplace(key, std::make_pair(g.jacobi_tile, g.jacobi_lz)).first;
        }
        if (it != tuned.end()) {
            g.jacobi_tile = it->second.first;
            g.jacobi_lz = it->second.second;
        }
    }
    if (zin_is_zero) {
        if (phases == 4) launch_tb<4, kRbgsZero>(st, g, zin, zout, rhs, 0, nz, 0, nz, 0, 0, kpar);
        else launch_tb<2, kRbgsZero>(st, g, zin, zout, rhs, 0, nz, 0, nz, 0, 0, kpar);
    } else {
        if (phases == 4) launch_tb<4, kRbgs>(st, g, zin, zout, rhs, 0, nz, 0, nz, 0, 0, kpar);
        else launch_tb<2, kRbgs>(st, g, zin, zout, rhs, 0, nz, 0, nz, 0, 0, kpar);
    }
    return true;
}

int jacobi_launches(const FsGrid& g, int sweeps, int naive)
{
    if (naive || !can_block(g)) return sweeps;
    if (g.jacobi_deep && can_deep(g) && g.wall_lo && g.wall_hi) {
        int deep, four, single;
        deep_split(sweeps, &deep, &four, &single);
        return deep + four + single;
    }
    return sweeps / kT + sweeps % kT;
}

int jacobi_deep_passes(const FsGrid& g, int sweeps, int naive)
{
    if (naive || !(g.jacobi_deep && can_deep(g) && g.wall_lo && g.wall_hi)) return 0;
    int deep, four, single;
    deep_split(sweeps, &deep, &four, &single);
    return deep;
}

// Single-slab driver: both z faces of the slab must be global walls.
void jacobi_multi(hipStream_t st, const FsGrid& g, float* p, float* np, const float* div,
                  int sweeps, int naive, int* result_in_np, int* prev_held)
{
    float *a = p, *b = np;
    int left = sweeps;
    if (!naive && g.jacobi_deep && can_deep(g) && g.wall_lo && g.wall_hi) {
        int deep, four, single;
        deep_split(sweeps, &deep, &four, &single);
        for (int q = 0; q < deep; ++q) {
            launch_deep(st, g, a, b, div);
            float* t = a; a = b; b = t;
        }
        left -= deep * kTDeep;
    }
    *prev_held = sweeps > 0 && (naive || !can_block(g) || left % kT != 0);
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
