// fs_advect.hip -- semi-Lagrangian advection for gfx950.
//
// Reference expressions restated here (CuSolver/Source/Solver.cu): zero-outside point sample
// :37-42, trilinear gather :52-113, RK4 :139-161, MacCormack :163-169, SemiLagKernel :370-403,
// Advect :1145-1165.  Arithmetic is binary32 in the reference's association order without
// FMA contraction.
//
// One output element costs 24 velocity gathers (MacCormack = 2 x RK4 = 8 stages x 3 staggered
// components) and one field gather, 8 taps each.  Measured on MI355X the limiter is not HBM but
// the CU's vector-memory address path (a wave-wide dword load holds it for ~4-7 cycles, a
// dwordx2 for ~15) and then the VALU.  So the velocity gathers are served from LDS: a workgroup
// stages the zero-extended velocity brick around its outputs (halo 2-3 cells) once, and every
// gather whose 8 taps fall inside the brick reads LDS (ds_read2_b32 pairs, ~4 cycles per pair);
// the rare gather that leaves the brick takes the global path below, which returns the same
// values.  The three velocity components are advected by one kernel so they share the brick.
#include "fs_kernels.h"

#include <climits>
#include <cstdlib>

#pragma clang fp contract(off)

namespace {

__host__ __device__ inline int cdiv(int a, int b) { return (a + b - 1) / b; }

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

// The two meanings of the reference's a*b + c shaped expressions.  Plain: every product and sum
// rounded on its own, the meaning of the source text and this library's default.  Fused: what a
// compiler that contracts (nvcc's default -fmad=true, which built the reference's binary) makes of
// them: the left product of lerp's sum of two products and the product of an RK position update fused
// into the add (FS_FLAG_FMAD).
struct Plain {
    static __device__ __forceinline__ float lerp(float a, float b, float s) { return a * (1.0f - s) + b * s; }   // Math.cuh:57-60
    static __device__ __forceinline__ float step(float p, float c, float u) { return p - c * u; }               // Solver.cu:145
};
struct Fused {
    static __device__ __forceinline__ float lerp(float a, float b, float s) { return __builtin_fmaf(a, 1.0f - s, b * s); }
    static __device__ __forceinline__ float step(float p, float c, float u) { return __builtin_fmaf(-c, u, p); }
};

// Solver.cu:109-112
template <class AR>
__device__ __forceinline__ float trilerp8(float a, float b, float c, float d, float e, float f,
                                          float g, float h, float fx, float fy, float fz)
{
    float l1 = AR::lerp(AR::lerp(a, b, fx), AR::lerp(c, d, fx), fy);
    float l2 = AR::lerp(AR::lerp(e, f, fx), AR::lerp(g, h, fx), fy);
    return AR::lerp(l1, l2, fz);
}

__device__ __forceinline__ int clampi(int v, int lo, int hi)
{
    return min(max(v, lo), hi);
}

// Limits of the lower tap index along z for a gather on this slab.
struct ZRange {
    int lo, hi;          // clamp range of the lower tap
    int chk_lo, chk_hi;  // a lower tap outside [chk_lo, chk_hi] left the ghost planes
};

// One gathered array on the global path: a raw buffer resource whose base is the lowest
// element any clamped tap can reach, and the constant that turns (ix, iy, iz) into a
// non-negative byte offset from it.
struct Fld {
    __amdgpu_buffer_rsrc_t rs;
    unsigned c;
    bool wide;   // plane * 4 does not fit a signed 24-bit multiply operand
};

__device__ __forceinline__ Fld make_fld(const float* f, const ZRange& zr, int pitch, int plane)
{
    // f is pre-offset so that global z indexes it directly; clamped taps reach
    // x, y in [-2, extent + 1] and z in [zr.lo, zr.hi + 1]
    long low = (long)zr.lo * plane - 2 * pitch - 2;
    unsigned bytes = ((unsigned)(zr.hi + 2 - zr.lo) * (unsigned)plane + 2u * pitch + 4u) * 4u;
    Fld F;
    F.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(f + low), 0, bytes, 0x00020000);
    F.c = 8u + 8u * (unsigned)pitch - 4u * (unsigned)zr.lo * (unsigned)plane;
    F.wide = plane >= (1 << 21);
    return F;
}

__device__ __forceinline__ float bload(__amdgpu_buffer_rsrc_t rs, unsigned voff, int soff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, soff, 0));
}

// Zero-outside trilinear gather (Solver.cu:52-113 with b_minmax_sampe = false) from global
// memory on the padded layout: clamping the lower tap to [-2, extent] lands every outside tap on
// a stored zero.  Eight dword buffer loads share one per-lane offset (row / plane steps ride in
// the scalar offset); the x+1 taps get their own offset register, hidden from the load combiner,
// because a dwordx2 costs the address path ~2x two dwords.
template <class AR>
__device__ __forceinline__ float gather_global(const Fld& F, Axis ax, Axis ay, Axis az, int dx,
                                               int dy, const ZRange& zr, int pitch, int plane)
{
    int ix = clampi(ax.i, -2, dx);
    int iy = clampi(ay.i, -2, dy);
    int iz = clampi(az.i, zr.lo, zr.hi);
    const int sp = pitch * 4, sq = plane * 4;
    // 24-bit multiplies issue at full rate (v_mul_lo_u32 takes four passes); the row term always
    // fits, the plane term fits for planes below 2^21 elements (the host sets F.wide otherwise)
    unsigned vo = F.c + (unsigned)(ix * 4) + (unsigned)__mul24(iy, sp) +
                  (unsigned)(F.wide ? iz * sq : __mul24(iz, sq));
    unsigned vo1 = vo + 4u;
    asm volatile("" : "+v"(vo1));
    float a = bload(F.rs, vo, 0);
    float b = bload(F.rs, vo1, 0);
    float c = bload(F.rs, vo, sp);
    float d = bload(F.rs, vo1, sp);
    float e = bload(F.rs, vo, sq);
    float f = bload(F.rs, vo1, sq);
    float g = bload(F.rs, vo, sq + sp);
    float h = bload(F.rs, vo1, sq + sp);
    return trilerp8<AR>(a, b, c, d, e, f, g, h, ax.f, ay.f, az.f);
}

// The velocity field as the kernels receive it (pointers pre-offset for global z).
struct Vel {
    const float* __restrict__ ux;
    const float* __restrict__ uy;
    const float* __restrict__ uz;
    int nx, ny, nz;
    int pitch, plane;
    ZRange zc;   // z limits for cell-centred-in-z arrays (ux, uy: dz = nz)
    ZRange zf;   // z limits for uz (dz = nz + 1)
};

// ---- samplers: "velocity at a point" = the three staggered gathers of one RK stage
// (Solver.cu:142-144) --------------------------------------------------------------------

template <class AR>
struct GlobalSampler {
    typedef AR Arith;
    static constexpr bool kHasGridStage = false;
    Fld ux, uy, uz;
    int nx, ny, pitch, plane;
    ZRange zc, zf;

    __device__ __forceinline__ void init(const Vel& v)
    {
        ux = make_fld(v.ux, v.zc, v.pitch, v.plane);
        uy = make_fld(v.uy, v.zc, v.pitch, v.plane);
        uz = make_fld(v.uz, v.zf, v.pitch, v.plane);
        nx = v.nx; ny = v.ny; pitch = v.pitch; plane = v.plane;
        zc = v.zc; zf = v.zf;
    }
    template <bool HALO>
    __device__ __forceinline__ float3 vel(float3 p, int& bad) const
    {
        Axis x0 = split_axis(p.x - 0.f), xh = split_axis(p.x - 0.5f);
        Axis y0 = split_axis(p.y - 0.f), yh = split_axis(p.y - 0.5f);
        Axis z0 = split_axis(p.z - 0.f), zh = split_axis(p.z - 0.5f);
        if (HALO)
            bad |= (zh.i < zc.chk_lo) | (zh.i > zc.chk_hi) | (z0.i < zf.chk_lo) | (z0.i > zf.chk_hi);
        float3 u;
        u.x = gather_global<AR>(ux, x0, yh, zh, nx + 1, ny, zc, pitch, plane);
        u.y = gather_global<AR>(uy, xh, y0, zh, nx, ny + 1, zc, pitch, plane);
        u.z = gather_global<AR>(uz, xh, yh, z0, nx, ny, zf, pitch, plane);
        return u;
    }
};

// LDS brick of the three velocity arrays around an output brick of BX x BY x BZ cells.
// All three arrays use the same index box [org, org + L) per axis; cell (x,y,z) of an array
// sits at lds[(x-org.x)*SX + (y-org.y)*SY + (z-org.z)*SZ].  The image is the zero-extended
// array: entries outside the array's own extent are 0, like the reference's sample outside.
// HXL / HXR: cells below / above the brick along x, HY and HZ on both sides of y and z; the +1 is
// the upper tap of the topmost lower-tap index.
template <int BX, int BY, int BZ, int HXL_, int HXR_, int HY_, int HZ_>
struct BrickDims {
    static_assert(BX % 4 == 0 && (HXL_ + 1) % 4 == 0, "brick rows are staged as aligned 16-byte quads");
    static constexpr int HXL = HXL_, HXR = HXR_, HY = HY_, HZ = HZ_;
    static constexpr int LX = BX + HXL + HXR + 1, LY = BY + 2 * HY + 1, LZ = BZ + 2 * HZ + 1;
    static constexpr int N = LX * LY * LZ;
    // Element (lx, ly, lz) sits at lx*SX + ly*SY + lz*SZ with z fastest and x slowest: the eight
    // taps of a gather then span SX + SY + SZ = 127 elements (narrow brick), so one address
    // register plus the 8-bit offsets of ds_read2_b32 reaches all of them, and lanes that differ
    // in x (the usual case within a wave) are SX elements apart, an odd number: no bank conflicts.
    static constexpr int SX = LY * LZ, SY = LZ, SZ = 1;
    static_assert(SX % 2 == 1, "x stride must be odd to spread lanes over the LDS banks");
};

// One coordinate split on the fast path: integer part and fraction.
struct SplitL {
    float f;
    int i;
};

template <class D, class AR>
struct BrickSampler {
    typedef AR Arith;
    static constexpr bool kHasGridStage = true;
    GlobalSampler<AR> G;
    const float* lds;    // 3 * D::N floats: ux, uy, uz
    int ox, oy, oz;      // global index of LDS element 0 along each axis
    float lo_x, lo_y, lo_z, hi_x, hi_y, hi_z;   // fast-path bounds, see vel()
    __device__ __forceinline__ void set_origin(int x, int y, int z)
    {
        ox = x; oy = y; oz = z;
        lo_x = (float)max(x, 0); lo_y = (float)max(y, 0); lo_z = (float)max(z, 0);
        hi_x = (float)(x + D::LX - 1); hi_y = (float)(y + D::LY - 1); hi_z = (float)(z + D::LZ - 1);
    }

    // ---- general path: literal split, per-gather choice between LDS and global memory ----
    __device__ __forceinline__ float gather(int which, const Fld& F, Axis ax, Axis ay, Axis az,
                                            int dx, int dy, const ZRange& zr) const
    {
        int lx = ax.i - ox, ly = ay.i - oy, lz = az.i - oz;
        bool in = ((unsigned)lx <= (unsigned)(D::LX - 2)) & ((unsigned)ly <= (unsigned)(D::LY - 2)) &
                  ((unsigned)lz <= (unsigned)(D::LZ - 2));
        if (__builtin_expect(in, 1)) {
            const float* q = lds + which * D::N + __mul24(lx, D::SX) + __mul24(ly, D::SY) + lz * D::SZ;
            float a = q[0], b = q[D::SX], c = q[D::SY], d = q[D::SY + D::SX];
            const float* q2 = q + D::SZ;
            float e = q2[0], f = q2[D::SX], g = q2[D::SY], h = q2[D::SY + D::SX];
            return trilerp8<AR>(a, b, c, d, e, f, g, h, ax.f, ay.f, az.f);
        }
        return gather_global<AR>(F, ax, ay, az, dx, dy, zr, G.pitch, G.plane);
    }
    template <bool HALO>
    __device__ __forceinline__ float3 vel_general(float3 p, int& bad) const
    {
        Axis x0 = split_axis(p.x - 0.f), xh = split_axis(p.x - 0.5f);
        Axis y0 = split_axis(p.y - 0.f), yh = split_axis(p.y - 0.5f);
        Axis z0 = split_axis(p.z - 0.f), zh = split_axis(p.z - 0.5f);
        if (HALO)
            bad |= (zh.i < G.zc.chk_lo) | (zh.i > G.zc.chk_hi) | (z0.i < G.zf.chk_lo) |
                   (z0.i > G.zf.chk_hi);
        float3 u;
        u.x = gather(0, G.ux, x0, yh, zh, G.nx + 1, G.ny, G.zc);
        u.y = gather(1, G.uy, xh, y0, zh, G.nx, G.ny + 1, G.zc);
        u.z = gather(2, G.uz, xh, yh, z0, G.nx, G.ny, G.zf);
        return u;
    }

    // ---- fast path, taken when it holds for every lane of the wave ----
    // For a coordinate c >= 0 the reference split (Solver.cu:58-82) is ix = floor(c),
    // fx = c - floor(c): truncation equals floor, the c < -1e-6 fix-up cannot fire, and
    // c - floor(c) is exact, which is what v_fract_f32 returns for c >= 0.
    static __device__ __forceinline__ SplitL split_fast(float c)
    {
        SplitL s;
        s.f = __builtin_amdgcn_fractf(c);
        s.i = (int)c;
        return s;
    }
    // cb = byte offset of the array inside the LDS image minus that of the brick origin: the
    // address of the lowest tap is three multiply-adds away from the integer parts
    __device__ __forceinline__ float gather_fast(int cb, const SplitL& sx, const SplitL& sy,
                                                 const SplitL& sz) const
    {
        int off = __mul24(sx.i, 4 * D::SX) + (__mul24(sy.i, 4 * D::SY) + ((sz.i << 2) + cb));
        // keep the sum in one register: the eight taps are then immediates off one address
        asm volatile("" : "+v"(off));
        const float* q = reinterpret_cast<const float*>(reinterpret_cast<const char*>(lds) + off);
        float a = q[0], b = q[D::SX], c = q[D::SY], d = q[D::SY + D::SX];
        const float* q2 = q + D::SZ;
        float e = q2[0], f = q2[D::SX], g = q2[D::SY], h = q2[D::SY + D::SX];
        return trilerp8<AR>(a, b, c, d, e, f, g, h, sx.f, sy.f, sz.f);
    }
    template <bool HALO>
    __device__ __forceinline__ float3 vel(float3 p, int& bad) const
    {
        const float xh = p.x - 0.5f, yh = p.y - 0.5f, zh = p.z - 0.5f;
        // every tap of every gather lies inside the brick, and every coordinate is >= 0, iff the
        // shifted coordinates are >= max(origin, 0) and the unshifted ones < origin + L - 1
        // (floor(c) + 1 <= origin + L - 1): six compares against wave-uniform bounds
        const bool ok = (xh >= lo_x) & (yh >= lo_y) & (zh >= lo_z) & (p.x < hi_x) & (p.y < hi_y) & (p.z < hi_z);
        // (the ballot of the complement is a scalar test of the compare masks: no vector op)
        if (__builtin_amdgcn_ballot_w64(!ok) != 0) return vel_general<HALO>(p, bad);
        const SplitL x0 = split_fast(p.x), x5 = split_fast(xh);
        const SplitL y0 = split_fast(p.y), y5 = split_fast(yh);
        const SplitL z0 = split_fast(p.z), z5 = split_fast(zh);
        if (HALO) bad |= (z5.i < G.zc.chk_lo) | (z5.i > G.zc.chk_hi) | (z0.i < G.zf.chk_lo) | (z0.i > G.zf.chk_hi);
        const int cb = -4 * (D::SX * ox + D::SY * oy + D::SZ * oz);
        float3 u;
        u.x = gather_fast(cb, x0, y5, z5);
        u.y = gather_fast(cb + 4 * D::N, x5, y0, z5);
        u.z = gather_fast(cb + 8 * D::N, x5, y5, z0);
        return u;
    }

    // ---- velocity at a sample point of the grid (first stage of the first RK4) ----
    // At a grid sample point every gather coordinate is an integer (weight 0) or an integer
    // plus one half (weight .5).  A weight-0 lerp returns its first operand for finite data
    // (a*1 + b*0, fused or not), so those taps are not read; the weight-.5 lerps run through lerp as usual.
    // MX/MY/MZ: the axis has weight .5.  q addresses the lower taps.
    template <bool MX, bool MY, bool MZ>
    static __device__ __forceinline__ float gather_grid(const float* q)
    {
        auto X = [](const float* r) { return MX ? AR::lerp(r[0], r[D::SX], 0.5f) : r[0]; };
        float v00 = X(q);
        float l1 = MY ? AR::lerp(v00, X(q + D::SY), 0.5f) : v00;
        if (!MZ) return l1;
        float v01 = X(q + D::SZ);
        float l2 = MY ? AR::lerp(v01, X(q + D::SZ + D::SY), 0.5f) : v01;
        return AR::lerp(l1, l2, 0.5f);
    }
    // DIR = 0 cell centre, 1/2/3 the x/y/z face of cell (i,j,k); the point is inside the brick.
    template <int DIR>
    __device__ __forceinline__ float3 vel_grid(int i, int j, int k) const
    {
        // sample offsets of the point: 0 on the face axis, .5 elsewhere (Solver.cu:381-383)
        constexpr bool px = DIR != 1, py = DIR != 2, pz = DIR != 3;   // true = offset .5
        const float* c = lds + (i - ox) * D::SX + (j - oy) * D::SY + (k - oz) * D::SZ;
        // array a has offset 0 on its own axis and .5 on the others; on an axis where point and
        // array offsets agree the coordinate is the integer index (weight 0); where the point has
        // 0 and the array .5 it is index - .5 (lower tap index-1); where the point has .5 and the
        // array 0 it is index + .5 (lower tap index)
        float3 u;
        {   // ux: array offsets (0, .5, .5)
            constexpr bool mx = px, my = !py, mz = !pz;
            const float* q = c - (my ? D::SY : 0) - (mz ? D::SZ : 0);
            u.x = gather_grid<mx, my, mz>(q);
        }
        {   // uy: (.5, 0, .5)
            constexpr bool mx = !px, my = py, mz = !pz;
            const float* q = c + D::N - (mx ? D::SX : 0) - (mz ? D::SZ : 0);
            u.y = gather_grid<mx, my, mz>(q);
        }
        {   // uz: (.5, .5, 0)
            constexpr bool mx = !px, my = !py, mz = pz;
            const float* q = c + 2 * D::N - (mx ? D::SX : 0) - (my ? D::SY : 0);
            u.z = gather_grid<mx, my, mz>(q);
        }
        return u;
    }
};

// Cooperative fill of the LDS brick: one array, zero outside its data box.
template <class D, int THREADS>
__device__ __forceinline__ void stage_array(float* dst, const float* __restrict__ src, int ox, int oy,
                                            int oz, int dx, int dy, int zd_lo, int zd_hi, int pitch,
                                            int plane)
{
    // Rows are read as aligned 16-byte quads (the brick's x origin is one element past a
    // 16-byte boundary, so quad 0 starts at lx = -1 and that element is dropped).  Raw buffer
    // over the array's data box: a quad of a row outside the box gets an offset past the end of
    // the buffer, for which the hardware returns 0; elements left or right of the array's x
    // extent are zeroed one by one.  All loads are issued before the first LDS write.
    typedef float f4 __attribute__((ext_vector_type(4)));
    const float* base = src + (long)zd_lo * plane;
    const unsigned bytes = zd_hi > zd_lo ? (unsigned)(zd_hi - zd_lo) * (unsigned)plane * 4u : 0u;
    __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, bytes, 0x00020000);
    constexpr int Q = (D::LX + 4) / 4;            // quads per row: lx = -1 .. 4Q-2
    constexpr int NQ = Q * D::LY * D::LZ;
    constexpr int PER = (NQ + THREADS - 1) / THREADS;
    f4 vals[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        int e = threadIdx.x + u * THREADS;
        int r = e / Q;
        int q = e - r * Q;
        int lz = r / D::LY;
        int ly = r - lz * D::LY;
        int x = ox - 1 + 4 * q, y = oy + ly, z = oz + lz;
        bool ok = ((unsigned)y < (unsigned)dy) & (z >= zd_lo) & (z < zd_hi) & (e < NQ);
        unsigned off = (unsigned)(x + y * pitch + (z - zd_lo) * plane) * 4u;
        f4 v = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(ok ? off : 0xfffffff0u), 0, 0));
        v.x = (unsigned)(x + 0) < (unsigned)dx ? v.x : 0.f;
        v.y = (unsigned)(x + 1) < (unsigned)dx ? v.y : 0.f;
        v.z = (unsigned)(x + 2) < (unsigned)dx ? v.z : 0.f;
        v.w = (unsigned)(x + 3) < (unsigned)dx ? v.w : 0.f;
        vals[u] = v;
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        int e = threadIdx.x + u * THREADS;
        if (e < NQ) {
            int r = e / Q;
            int q = e - r * Q;
            const int lz = r / D::LY, ly = r - lz * D::LY;
            float* d = dst + (4 * q - 1) * D::SX + ly * D::SY + lz * D::SZ;    // element lx = 4q - 1
            if (q > 0) d[0] = vals[u].x;
            d[D::SX] = vals[u].y;
            d[2 * D::SX] = vals[u].z;
            if (4 * q + 2 < D::LX) d[3 * D::SX] = vals[u].w;
        }
    }
}

template <class D, int THREADS>
__device__ __forceinline__ void stage_velocity(float* lds, const Vel& v, int ox, int oy, int oz)
{
    // data boxes: planes that hold this rank's copy of each array (ghost planes included)
    int zc_lo = max(v.zc.lo, 0), zc_hi = min(v.zc.hi + 2, v.nz);
    int zf_lo = max(v.zf.lo, 0), zf_hi = min(v.zf.hi + 2, v.nz + 1);
    stage_array<D, THREADS>(lds, v.ux, ox, oy, oz, v.nx + 1, v.ny, zc_lo, zc_hi, v.pitch, v.plane);
    stage_array<D, THREADS>(lds + D::N, v.uy, ox, oy, oz, v.nx, v.ny + 1, zc_lo, zc_hi, v.pitch, v.plane);
    stage_array<D, THREADS>(lds + 2 * D::N, v.uz, ox, oy, oz, v.nx, v.ny, zf_lo, zf_hi, v.pitch, v.plane);
}

// ---- back-trace ----------------------------------------------------------------------------

// Solver.cu:139-161.  (0.5f*dt) is formed first, then scales each component
// (Math.cuh:80-86); the final update is  pos - (dt * sum) / 6  with a true division (fsk::div6
// returns the same correctly rounded quotient).
template <bool HALO, class S>
__device__ __forceinline__ float3 rk4(const S& s, float3 pos, float dt, int& bad, const float3* first)
{
    typedef typename S::Arith AR;
    float hdt = 0.5f * dt;
    float3 u1 = first ? *first : s.template vel<HALO>(pos, bad);
    float3 p1 = make_float3(AR::step(pos.x, hdt, u1.x), AR::step(pos.y, hdt, u1.y), AR::step(pos.z, hdt, u1.z));
    float3 u2 = s.template vel<HALO>(p1, bad);
    float3 p2 = make_float3(AR::step(pos.x, hdt, u2.x), AR::step(pos.y, hdt, u2.y), AR::step(pos.z, hdt, u2.z));
    float3 u3 = s.template vel<HALO>(p2, bad);
    float3 p3 = make_float3(AR::step(pos.x, dt, u3.x), AR::step(pos.y, dt, u3.y), AR::step(pos.z, dt, u3.z));
    float3 u4 = s.template vel<HALO>(p3, bad);
    float sx = ((u1.x + 2.0f * u2.x) + 2.0f * u3.x) + u4.x;
    float sy = ((u1.y + 2.0f * u2.y) + 2.0f * u3.y) + u4.y;
    float sz = ((u1.z + 2.0f * u2.z) + 2.0f * u3.z) + u4.z;
    return make_float3(pos.x - fsk::div6(dt * sx), pos.y - fsk::div6(dt * sy), pos.z - fsk::div6(dt * sz));
}

// Solver.cu:163-169 (MAC) or the plain RK4 trace of Solver.cu:396.
// `first`: velocity at pos when the caller already has it (pos is a grid sample point).
template <bool MAC, bool HALO, class S>
__device__ __forceinline__ float3 backtrace(const S& s, float3 pos, float dt, int& bad,
                                            const float3* first = nullptr)
{
    float3 pred = rk4<HALO>(s, pos, dt, bad, first);
    if (!MAC) return pred;
    float3 refl = rk4<HALO>(s, pred, -dt, bad, nullptr);
    return make_float3(pred.x + (pos.x - refl.x) / 2.f, pred.y + (pos.y - refl.y) / 2.f,
                       pred.z + (pos.z - refl.z) / 2.f);
}

// One advected velocity component at face (i,j,k): Solver.cu:377-400 for dir = DIR.
template <int DIR, bool MAC, bool HALO, class S>
__device__ __forceinline__ float advect_component(const S& s, const GlobalSampler<typename S::Arith>& G, int i, int j,
                                                  int k, float dt, int& bad)
{
    float3 pos;   // Solver.cu:381-383
    pos.x = (float)i + (DIR == 1 ? 0.f : 0.5f);
    pos.y = (float)j + (DIR == 2 ? 0.f : 0.5f);
    pos.z = (float)k + (DIR == 3 ? 0.f : 0.5f);
    float3 c;
    if constexpr (S::kHasGridStage) {
        float3 u1 = s.template vel_grid<DIR>(i, j, k);
        c = backtrace<MAC, HALO>(s, pos, dt, bad, &u1);
    } else {
        c = backtrace<MAC, HALO>(s, pos, dt, bad);
    }
    Axis ax = split_axis(c.x - (DIR == 1 ? 0.f : 0.5f));
    Axis ay = split_axis(c.y - (DIR == 2 ? 0.f : 0.5f));
    Axis az = split_axis(c.z - (DIR == 3 ? 0.f : 0.5f));
    const ZRange& zr = DIR == 3 ? G.zf : G.zc;
    if (HALO) bad |= (az.i < zr.chk_lo) | (az.i > zr.chk_hi);
    // the advected field is the component itself (Solver.cu:1153-1155)
    return gather_global<typename S::Arith>(DIR == 1 ? G.ux : (DIR == 2 ? G.uy : G.uz), ax, ay, az,
                         DIR == 1 ? G.nx + 1 : G.nx, DIR == 2 ? G.ny + 1 : G.ny, zr, G.pitch,
                         G.plane);
}

// owned_only: the launch runs while the ghost planes are still being received, so a gather whose
// taps leave the planes this rank owns (cells [z0, z0+nzl), faces [z0, z0+nzl]) must raise the flag
__host__ inline ZRange make_zrange(const FsGrid& g, int dz, bool owned_only)
{
    ZRange r;
    r.lo = g.wall_lo ? -2 : g.z0 - g.ghost;
    r.hi = g.wall_hi ? dz : g.z0 + g.nzl + g.ghost - 1;
    r.chk_lo = g.wall_lo ? INT_MIN : (owned_only ? g.z0 : r.lo);
    r.chk_hi = g.wall_hi ? INT_MAX : (owned_only ? g.z0 + g.nzl - 2 + (dz - g.nz) : r.hi);
    return r;
}

__host__ inline Vel make_vel(const FsGrid& g, const float* ux, const float* uy, const float* uz,
                             bool owned_only = false)
{
    Vel v;
    ptrdiff_t zoff = (ptrdiff_t)g.z0 * g.plane;
    v.ux = ux - zoff;
    v.uy = uy - zoff;
    v.uz = uz - zoff;
    v.nx = g.nx;
    v.ny = g.ny;
    v.nz = g.nz;
    v.pitch = g.pitch;
    v.plane = g.plane;
    v.zc = make_zrange(g, g.nz, owned_only);
    v.zf = make_zrange(g, g.nz + 1, owned_only);
    return v;
}

// ---------------------------------------------------------------- brick mapping

// Workgroups are dealt round-robin to the 8 XCDs, so ids b and b+8 share an L2.  Give each
// XCD one contiguous run of bricks (bijective for any count).
__device__ __forceinline__ int xcd_contiguous(int b, int nb)
{
    int q = nb >> 3, r = nb & 7;
    int xcd = b & 7, loc = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

// The box of cells one launch covers and its decomposition into bricks.
struct Box {
    int x0, y0, z0;      // first cell (global indices)
    int nx, ny, nz;      // extent in cells
    int bx, by;          // bricks along x and y
    int nb;              // total bricks
};

__host__ inline Box make_box(int tx, int ty, int tz, int x0, int y0, int z0, int nx, int ny, int nz)
{
    Box b;
    b.x0 = x0; b.y0 = y0; b.z0 = z0;
    b.nx = nx; b.ny = ny; b.nz = nz;
    b.bx = cdiv(nx, tx);
    b.by = cdiv(ny, ty);
    b.nb = b.bx * b.by * cdiv(nz, tz);
    return b;
}

// brick coordinates (in bricks) of this workgroup
__device__ __forceinline__ void brick_of_block(const Box& B, int& bx, int& by, int& bz)
{
    int id = xcd_contiguous((int)blockIdx.x, B.nb);
    bz = id / (B.bx * B.by);
    int rem = id - bz * (B.bx * B.by);
    by = rem / B.bx;
    bx = rem - by * B.bx;
}

// ---------------------------------------------------------------- kernels

// Brick shape of the LDS kernels: 32 x 8 x 4 cells, halo 3, 512 threads (two cells each).
constexpr int kBX = 32, kBY = 8, kBZ = 4, kThreads = 512;
// Two halo widths.  The narrow brick (2 cells in y and z and above x, 3 below x, which the quad
// alignment fixes) takes 53 KB, so three workgroups share a CU (six waves per SIMD, <= 80 VGPRs);
// the wide one (3 cells everywhere) takes 77 KB, two workgroups per CU.  A back-trace stage that
// leaves the brick in any lane sends its wave through the general path, which is slower but
// returns the same values, so the choice only matters for speed: the narrow brick is 10-20 %
// faster as long as displacements stay below about 1.5 cells per stage (|u| dt), the wide one
// tolerates about 2.5.  FS_ADVECT_BRICK=0 selects the wide brick.
struct DimsWide : BrickDims<kBX, kBY, kBZ, 3, 3, 3, 3> { static constexpr int kWaves = 4; };
struct DimsNarrow : BrickDims<kBX, kBY, kBZ, 3, 2, 2, 2> { static constexpr int kWaves = 6; };

// ux, uy, uz advected from the old velocity, all faces with i < B.x0 + B.nx.  Cell (i,j,k) of
// the box owns face i of ux (if j < ny, k < zend), face j of uy (if k < zend) and face k of uz
// (if j < ny); the box spans j in [0, ny] and k in [z0, zend].
template <class Dims, bool MAC, bool HALO, class AR = Plain>
__global__ void __launch_bounds__(kThreads, Dims::kWaves)
k_advect_velocity_lds(float* __restrict__ nux, float* __restrict__ nuy, float* __restrict__ nuz, Vel v,
                      float dt, Box B, int zend, int* halo_flag)
{
    __shared__ float lds[3 * Dims::N];
    int bx, by, bz;
    brick_of_block(B, bx, by, bz);
    const int X0 = B.x0 + bx * kBX, Y0 = B.y0 + by * kBY, Z0 = B.z0 + bz * kBZ;
    BrickSampler<Dims, AR> S;
    S.G.init(v);
    S.lds = lds;
    S.set_origin(X0 - Dims::HXL, Y0 - Dims::HY, Z0 - Dims::HZ);
    stage_velocity<Dims, kThreads>(lds, v, S.ox, S.oy, S.oz);
    __syncthreads();

    const int t = threadIdx.x;
    const int lx = t % kBX, ly = (t / kBX) % kBY, lz0 = t / (kBX * kBY);
    int bad = 0;
#pragma unroll 1
    for (int lz = lz0; lz < kBZ; lz += kThreads / (kBX * kBY)) {
        const int i = X0 + lx, j = Y0 + ly, k = Z0 + lz;
        if (i >= B.x0 + B.nx || j >= B.y0 + B.ny || k >= B.z0 + B.nz) continue;
        const int o = i + j * v.pitch + k * v.plane;
        if (j < v.ny && k < zend) nux[o] = advect_component<1, MAC, HALO>(S, S.G, i, j, k, dt, bad);
        if (k < zend) nuy[o] = advect_component<2, MAC, HALO>(S, S.G, i, j, k, dt, bad);
        if (j < v.ny) nuz[o] = advect_component<3, MAC, HALO>(S, S.G, i, j, k, dt, bad);
    }
    if (HALO && bad) atomicOr(halo_flag, 1);
}

// T and rho advected with the new velocity; they share one back-trace (same sample position,
// Solver.cu:381-383 with dir = 0, and same velocity, Solver.cu:1160-1164).  With DIV the kernel
// also writes the divergence of that velocity (Solver.cu:406-423) for its cells, straight from
// the staged bricks: the projection that follows then needs no separate pass over the velocity.
template <class Dims, bool MAC, bool HALO, bool DIV, class AR = Plain>
__global__ void __launch_bounds__(kThreads, Dims::kWaves)
k_advect_scalars_lds(const float* __restrict__ T, float* __restrict__ nT, const float* __restrict__ rho,
                     float* __restrict__ nrho, float* __restrict__ div, Vel v, float dt, Box B,
                     int* halo_flag)
{
    __shared__ float lds[3 * Dims::N];
    int bx, by, bz;
    brick_of_block(B, bx, by, bz);
    const int X0 = B.x0 + bx * kBX, Y0 = B.y0 + by * kBY, Z0 = B.z0 + bz * kBZ;
    BrickSampler<Dims, AR> S;
    S.G.init(v);
    S.lds = lds;
    S.set_origin(X0 - Dims::HXL, Y0 - Dims::HY, Z0 - Dims::HZ);
    stage_velocity<Dims, kThreads>(lds, v, S.ox, S.oy, S.oz);
    __syncthreads();
    const Fld FT = make_fld(T, v.zc, v.pitch, v.plane), FR = make_fld(rho, v.zc, v.pitch, v.plane);

    const int t = threadIdx.x;
    const int lx = t % kBX, ly = (t / kBX) % kBY, lz0 = t / (kBX * kBY);
    int bad = 0;
#pragma unroll 1
    for (int lz = lz0; lz < kBZ; lz += kThreads / (kBX * kBY)) {
        const int i = X0 + lx, j = Y0 + ly, k = Z0 + lz;
        if (i >= B.x0 + B.nx || j >= B.y0 + B.ny || k >= B.z0 + B.nz) continue;
        float3 pos = make_float3((float)i + 0.5f, (float)j + 0.5f, (float)k + 0.5f);
        float3 u1 = S.template vel_grid<0>(i, j, k);
        float3 c = backtrace<MAC, HALO>(S, pos, dt, bad, &u1);
        Axis ax = split_axis(c.x - 0.5f), ay = split_axis(c.y - 0.5f), az = split_axis(c.z - 0.5f);
        if (HALO) bad |= (az.i < v.zc.chk_lo) | (az.i > v.zc.chk_hi);
        const int o = i + j * v.pitch + k * v.plane;
        nT[o] = gather_global<AR>(FT, ax, ay, az, v.nx, v.ny, v.zc, v.pitch, v.plane);
        nrho[o] = gather_global<AR>(FR, ax, ay, az, v.nx, v.ny, v.zc, v.pitch, v.plane);
        if (DIV) {
            const float* q = lds + (i - S.ox) * Dims::SX + (j - S.oy) * Dims::SY + (k - S.oz) * Dims::SZ;
            const float ul = q[0], ur = q[Dims::SX];
            const float ubo = q[Dims::N], ut = q[Dims::N + Dims::SY];
            const float ubh = q[2 * Dims::N], uf = q[2 * Dims::N + Dims::SZ];
            div[o] = ur + uf + ut - ul - ubh - ubo;          // Solver.cu:420, that order
        }
    }
    if (HALO && bad) atomicOr(halo_flag, 1);
}

// Global-path kernel for one velocity component over an arbitrary box of faces (used for the
// faces the brick kernel does not cover, and for A/B timing): TX x TY x TZ threads, one face each.
template <int DIR, bool MAC, bool HALO, int TX, int TY, int TZ, class AR = Plain>
__global__ void __launch_bounds__(TX * TY * TZ)
k_advect_component_global(float* __restrict__ out, Vel v, float dt, Box B, int* halo_flag)
{
    int bx, by, bz;
    brick_of_block(B, bx, by, bz);
    const int t = threadIdx.x;
    const int x = bx * TX + t % TX, y = by * TY + (t / TX) % TY, z = bz * TZ + t / (TX * TY);
    if (x >= B.nx || y >= B.ny || z >= B.nz) return;
    const int i = B.x0 + x, j = B.y0 + y, k = B.z0 + z;
    GlobalSampler<AR> G;
    G.init(v);
    int bad = 0;
    out[i + j * v.pitch + k * v.plane] = advect_component<DIR, MAC, HALO>(G, G, i, j, k, dt, bad);
    if (HALO && bad) atomicOr(halo_flag, 1);
}

template <bool MAC, bool HALO, int TX, int TY, int TZ, class AR = Plain>
__global__ void __launch_bounds__(TX * TY * TZ)
k_advect_scalars_global(const float* __restrict__ T, float* __restrict__ nT,
                        const float* __restrict__ rho, float* __restrict__ nrho, Vel v, float dt, Box B,
                        int* halo_flag)
{
    int bx, by, bz;
    brick_of_block(B, bx, by, bz);
    const int t = threadIdx.x;
    const int x = bx * TX + t % TX, y = by * TY + (t / TX) % TY, z = bz * TZ + t / (TX * TY);
    if (x >= B.nx || y >= B.ny || z >= B.nz) return;
    const int i = B.x0 + x, j = B.y0 + y, k = B.z0 + z;
    GlobalSampler<AR> G;
    G.init(v);
    const Fld FT = make_fld(T, v.zc, v.pitch, v.plane), FR = make_fld(rho, v.zc, v.pitch, v.plane);
    int bad = 0;
    float3 pos = make_float3((float)i + 0.5f, (float)j + 0.5f, (float)k + 0.5f);
    float3 c = backtrace<MAC, HALO>(G, pos, dt, bad);
    Axis ax = split_axis(c.x - 0.5f), ay = split_axis(c.y - 0.5f), az = split_axis(c.z - 0.5f);
    if (HALO) bad |= (az.i < v.zc.chk_lo) | (az.i > v.zc.chk_hi);
    const int o = i + j * v.pitch + k * v.plane;
    nT[o] = gather_global<AR>(FT, ax, ay, az, v.nx, v.ny, v.zc, v.pitch, v.plane);
    nrho[o] = gather_global<AR>(FR, ax, ay, az, v.nx, v.ny, v.zc, v.pitch, v.plane);
    if (HALO && bad) atomicOr(halo_flag, 1);
}

// FS_ADVECT_GLOBAL (timing experiments): bit 0 = velocity on the global path, bit 1 = scalars
// on the global path.  Default: both through LDS bricks.
bool narrow_brick(const FsGrid& g) { return g.env.advect_brick != 0; }
bool use_global_path(const FsGrid& g) { return (g.env.advect_global & 1) != 0; }
bool scalars_use_lds(const FsGrid& g) { return (g.env.advect_global & 2) == 0; }

// CALL(M, H, AR) with the run-time scheme, slab and arithmetic choices as template arguments
#define FS_DISPATCH_MH(mac, halo, fmad, CALL)      \
    do {                                           \
        if (fmad) {                                \
            if (mac && halo) CALL(true, true, Fused);        \
            else if (mac) CALL(true, false, Fused);          \
            else if (halo) CALL(false, true, Fused);         \
            else CALL(false, false, Fused);                  \
        } else {                                   \
            if (mac && halo) CALL(true, true, Plain);        \
            else if (mac) CALL(true, false, Plain);          \
            else if (halo) CALL(false, true, Plain);         \
            else CALL(false, false, Plain);                  \
        }                                          \
    } while (0)

template <int DIR>
void launch_component_global(hipStream_t st, float* out, const Vel& v, float dt, bool mac, bool halo, bool fmad,
                             int x0, int y0, int z0, int nx, int ny, int nz, bool thin_x,
                             int* halo_flag)
{
    if (nx <= 0 || ny <= 0 || nz <= 0) return;
    if (thin_x) {
        Box B = make_box(1, 64, 4, x0, y0, z0, nx, ny, nz);
#define FS_CALL(M, H, AR)                                                                          \
    hipLaunchKernelGGL((k_advect_component_global<DIR, M, H, 1, 64, 4, AR>), dim3(B.nb), dim3(256), 0, st, \
                       out, v, dt, B, halo_flag)
        FS_DISPATCH_MH(mac, halo, fmad, FS_CALL);
#undef FS_CALL
    } else {
        Box B = make_box(64, 4, 1, x0, y0, z0, nx, ny, nz);
#define FS_CALL(M, H, AR)                                                                          \
    hipLaunchKernelGGL((k_advect_component_global<DIR, M, H, 64, 4, 1, AR>), dim3(B.nb), dim3(256), 0, st, \
                       out, v, dt, B, halo_flag)
        FS_DISPATCH_MH(mac, halo, fmad, FS_CALL);
#undef FS_CALL
    }
}

}  // namespace

namespace fsk {

int advect_velocity_launches(const FsGrid& g) { return use_global_path(g) ? 3 : 2; }

void advect_velocity(hipStream_t st, const FsGrid& g, float* nux, float* nuy, float* nuz,
                     const float* ux, const float* uy, const float* uz, float dt, int scheme,
                     int* halo_flag, int zb, int ze, bool owned_only)
{
    const int zend = g.z0 + g.nzl;
    if (ze < 0) {   // default: everything this rank computes
        zb = g.z0;
        ze = zend + 1;
    }
    if (ze <= zb) return;
    Vel v = make_vel(g, ux, uy, uz, owned_only);
    const bool halo = !(g.wall_lo && g.wall_hi), mac = scheme != 0, fmad = g.fmad != 0;
    const ptrdiff_t zoff = (ptrdiff_t)g.z0 * g.plane;
    float *ox = nux - zoff, *oy = nuy - zoff, *oz = nuz - zoff;
    const int nzc = min(ze, zend) - zb;   // planes of [zb, ze) that hold ux / uy faces
    if (use_global_path(g)) {
        launch_component_global<1>(st, ox, v, dt, mac, halo, fmad, 0, 0, zb, g.nx + 1, g.ny, nzc, false, halo_flag);
        launch_component_global<2>(st, oy, v, dt, mac, halo, fmad, 0, 0, zb, g.nx, g.ny + 1, nzc, false, halo_flag);
        launch_component_global<3>(st, oz, v, dt, mac, halo, fmad, 0, 0, zb, g.nx, g.ny, ze - zb, false, halo_flag);
        return;
    }
    // brick kernel: all faces with i < nx; the ux faces at i = nx go through the global path
    Box B = make_box(kBX, kBY, kBZ, 0, 0, zb, g.nx, g.ny + 1, ze - zb);
#define FS_CALL(M, H, AR)                                                                          \
    do {                                                                                           \
        if (narrow_brick(g))                                                                       \
            hipLaunchKernelGGL((k_advect_velocity_lds<DimsNarrow, M, H, AR>), dim3(B.nb), dim3(kThreads), 0, st, \
                               ox, oy, oz, v, dt, B, zend, halo_flag);                             \
        else                                                                                       \
            hipLaunchKernelGGL((k_advect_velocity_lds<DimsWide, M, H, AR>), dim3(B.nb), dim3(kThreads), 0, st, \
                               ox, oy, oz, v, dt, B, zend, halo_flag);                             \
    } while (0)
    FS_DISPATCH_MH(mac, halo, fmad, FS_CALL);
#undef FS_CALL
    launch_component_global<1>(st, ox, v, dt, mac, halo, fmad, g.nx, 0, zb, 1, g.ny, nzc, true, halo_flag);
}

bool advect_scalars(hipStream_t st, const FsGrid& g, const float* T, float* nT, const float* rho,
                    float* nrho, const float* ux, const float* uy, const float* uz, float dt,
                    int scheme, int* halo_flag, float* div, int zb, int ze, bool owned_only)
{
    if (ze < 0) {   // default: all owned cell planes
        zb = g.z0;
        ze = g.z0 + g.nzl;
    }
    if (ze <= zb) return div != nullptr && scalars_use_lds(g);
    Vel v = make_vel(g, ux, uy, uz, owned_only);
    const bool halo = !(g.wall_lo && g.wall_hi), mac = scheme != 0, fmad = g.fmad != 0;
    const ptrdiff_t zoff = (ptrdiff_t)g.z0 * g.plane;
    if (!scalars_use_lds(g)) {
        Box B = make_box(64, 4, 1, 0, 0, zb, g.nx, g.ny, ze - zb);
#define FS_CALL(M, H, AR)                                                                         \
    hipLaunchKernelGGL((k_advect_scalars_global<M, H, 64, 4, 1, AR>), dim3(B.nb), dim3(256), 0, st, \
                       T - zoff, nT - zoff, rho - zoff, nrho - zoff, v, dt, B, halo_flag)
        FS_DISPATCH_MH(mac, halo, fmad, FS_CALL);
#undef FS_CALL
        return false;
    }
    Box B = make_box(kBX, kBY, kBZ, 0, 0, zb, g.nx, g.ny, ze - zb);
    float* dv = div ? div - zoff : nullptr;
#define FS_CALL2(D, M, H, DV, AR)                                                                 \
    hipLaunchKernelGGL((k_advect_scalars_lds<D, M, H, DV, AR>), dim3(B.nb), dim3(kThreads), 0, st, T - zoff, \
                       nT - zoff, rho - zoff, nrho - zoff, dv, v, dt, B, halo_flag)
#define FS_CALL(M, H, AR)                                                                         \
    do {                                                                                          \
        if (narrow_brick(g)) {                                                                    \
            if (dv) FS_CALL2(DimsNarrow, M, H, true, AR);                                         \
            else FS_CALL2(DimsNarrow, M, H, false, AR);                                           \
        } else {                                                                                  \
            if (dv) FS_CALL2(DimsWide, M, H, true, AR);                                           \
            else FS_CALL2(DimsWide, M, H, false, AR);                                             \
        }                                                                                         \
    } while (0)
    FS_DISPATCH_MH(mac, halo, fmad, FS_CALL);
#undef FS_CALL
#undef FS_CALL2
    return dv != nullptr;
}

}  // namespace fsk
