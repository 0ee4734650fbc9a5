// fs_kernels.h -- launch interface between the solver object and the gfx950 kernels.
//
// Device layout shared by every field of a solver (one slab per rank):
//   element (x, y, z_global) lives at  base[x + y*pitch + (z_global - z0)*plane]
// with pitch >= nx + 3 and plane = pitch * (ny + 3).  The floats of a row beyond the
// field's own x extent, the rows beyond its y extent and the planes beyond its z
// extent at a global wall are zero and are never written by any kernel.  Because rows
// and planes are contiguous, x = -2,-1 of a row alias the zero tail of the previous row
// and y = -2,-1 of a plane alias the zero tail rows of the previous plane, so a gather
// whose indices are clamped to [-2, extent] reads 0 outside the field without a single
// bounds test (that is the reference's zero-outside sample, Solver.cu:37-42).
#pragma once
#include <hip/hip_runtime.h>

// Environment switches of the kernels' launchers (experiments and tests; the defaults are the
// measured best).  Read once per solver, at creation (fsk::read_env), never per launch.
struct FsEnv {
    int jacobi_cfg = -1;          // FS_JACOBI_CFG    tile variant 0..3 of the blocked Jacobi kernel, -1 = timed / by rule
    int jacobi_nt = -1;           // FS_JACOBI_NT     1 / 0 force / forbid non-temporal stores, -1 = timed
    int jacobi_order = 1;         // FS_JACOBI_ORDER  1 = XCD-contiguous workgroup order
    int jacobi_lz = 0;            // FS_JACOBI_LZ     planes per z chunk, 0 = timed / cost model
    int jacobi_tune = 1;          // FS_JACOBI_TUNE   0 = no timing at creation, 2 = print the readings
    int jacobi_deep = -1;         // FS_JACOBI_DEEP   1 / 0 force / forbid the six-sweep passes, -1 = timed
    int jacobi_deep_lz = 0;       // FS_JACOBI_DEEP_LZ  planes per z chunk of a six-sweep pass, 0 = timed / cost model
    int jacobi_deep_nt = -1;      // FS_JACOBI_DEEP_NT  its store policy, -1 = timed
    int advect_global = 0;        // FS_ADVECT_GLOBAL bit 0 / 1: velocity / scalar advection without LDS bricks
    int advect_brick = 1;         // FS_ADVECT_BRICK  1 = narrow LDS brick, 0 = wide
    long mg_tb_min = 128L * 128L; // FS_MG_TB_MIN     smallest plane (cells) smoothed by the blocked kernel
    int mg_lds_pad = 1;           // FS_MG_LDS_PAD    0 = the older single-workgroup smoother
    int mg_no_lds = 0;            // FS_MG_NO_LDS     1 = small levels one launch per colour
    int mg_slab_ghost = 0;        // FS_MG_SLAB_GHOST cap on the ghost depth of a slab's multigrid levels, 0 = none
    int vort_lattice = 1;         // FS_VORT_LATTICE  0 = every confined face evaluates its own |omega|
    int slab_pghost = 8;          // FS_SLAB_PGHOST   4 = one exchange per fused Jacobi pass
    int xchg_streams = -1;        // FS_XCHG_STREAMS  1 / 0 force the two-stream / one-stream ordering
    int xchg_debug = 0;           // FS_DEBUG_XCHG
    int xchg_host_wait = -1;      // FS_XCHG_HOST_WAIT
    int xchg_timeout_ms = 60000;  // FS_XCHG_TIMEOUT_MS
};

struct FsGrid {
    int nx, ny, nz;   // global cell counts
    int z0;           // global index of the first owned cell plane
    int nzl;          // owned cell planes on this rank
    int nz_min;       // owned cell planes of the thinnest slab of the decomposition (= nz on a single domain)
    int ghost;        // planes below z0 and above the last owned face plane an advection gather may read
    int pghost;       // ghost planes held per side (>= ghost): what the pressure, its divergence and the
                      // velocity behind that divergence use on a slab
    int pitch;        // floats per row
    int plane;        // floats per z plane
    int wall_lo;      // 1 if z0 is the global z = 0 wall
    int wall_hi;      // 1 if z0 + nzl is the global z = nz wall
    int jacobi_tile;  // 1 + tile variant of the blocked Jacobi kernel chosen for this grid, 0 = by rule
    int jacobi_lz;    // planes per z chunk of a blocked pass chosen for this grid, 0 = by the cost model
    int jacobi_nt;    // 1 = the blocked Jacobi passes of this grid store with the non-temporal policy
    int jacobi_deep;      // 1 = the sweeps run in passes of six where they can (on a z-slab: the same on every rank)
    int jacobi_deep_lz;   // planes per z chunk of a six-sweep pass, 0 = by the cost model
    int jacobi_deep_nt;   // 1 = the six-sweep passes store with the non-temporal policy
    int fmad;         // 1 = FS_FLAG_FMAD: the advection, the buoyancy and the confinement gathers fuse their
                      // multiply-adds the way a contracting compiler does (see fs_advect.hip)
    FsEnv env;        // the environment switches as they stood when the solver was created
};

struct FsFields {
    float *ux, *uy, *uz, *nux, *nuy, *nuz;
    float *rho, *nrho, *T, *nT;
    float *p, *np, *div;
};

struct FsSource {
    float x, y, z, radius, rho0, T0;
};

namespace fsk {

FsEnv read_env();   // the FS_* switches of the process environment

// contiguous float ranges copied by one launch (either side may be pinned host memory)
constexpr int kCopyBatchMax = 10;
struct CopyBatch {
    float* dst[kCopyBatchMax];
    const float* src[kCopyBatchMax];
    size_t n[kCopyBatchMax];
    int count;
};
void copy_batch(hipStream_t st, const CopyBatch& b);

#if defined(__HIPCC__)
// x / 6.f, correctly rounded for every binary32 x (what the true division of the reference yields,
// Solver.cu:160, 441), as RN32(RN64(x * RN64(1/6))): three instructions instead of the ten of the
// IEEE division sequence.  Why the two roundings cannot disagree with the one of the exact
// quotient: the product is within 2^-52 (relative) of x/6; x/6 with a 24-bit x is either exactly
// a binary32 rounding boundary -- only possible for a subnormal quotient, and then the product
// rounds to that boundary itself, so the narrowing conversion breaks the same tie to even -- or at
// least 2^-27 (relative) away from every boundary, because 6 * boundary has an odd factor that x
// cannot match.  Zeros keep their sign, infinities and NaNs pass through.
__device__ __forceinline__ float div6(float x)
{
    return (float)((double)x * 0x1.5555555555555p-3);
}
#endif

void fill_interior(hipStream_t st, const FsGrid& g, float* f, float value);
void source(hipStream_t st, const FsGrid& g, const FsSource& src, float* rho, float* T);
void force_uy(hipStream_t st, const FsGrid& g, float* uy, const float* T, float dt, float T_env);
// Advect all three velocity components from the old velocity (ux,uy,uz) into (nux,nuy,nuz).
// [zb, ze): the planes to produce (global indices within [z0, z0 + nzl], the last being the shared
// uz face plane); ze < 0 = all of them.  owned_only: the ghost planes are not valid yet, so a
// back-trace that needs one raises the halo flag instead of reading it.
void advect_velocity(hipStream_t st, const FsGrid& g, float* nux, float* nuy, float* nuz,
                     const float* ux, const float* uy, const float* uz, float dt, int scheme,
                     int* halo_flag, int zb = 0, int ze = -1, bool owned_only = false);
int advect_velocity_launches(const FsGrid& g);
// T and rho share one back-trace (same sample position, Solver.cu:381-383 with dir = 0).
// With div != nullptr the divergence of (ux,uy,uz) over the produced cells is written as well;
// returns whether it was.  [zb, ze) within [z0, z0 + nzl) and owned_only as above.
bool advect_scalars(hipStream_t st, const FsGrid& g, const float* T, float* nT, const float* rho,
                    float* nrho, const float* ux, const float* uy, const float* uz, float dt,
                    int scheme, int* halo_flag, float* div = nullptr, int zb = 0, int ze = -1,
                    bool owned_only = false);
// cell planes [zb, ze) (global indices); the faces uz[ze] must be held locally
void divergence(hipStream_t st, const FsGrid& g, float* div, const float* ux, const float* uy,
                const float* uz, int zb, int ze);
// one Jacobi sweep p -> np over the owned planes [zb, ze) (global indices)
void jacobi_sweep(hipStream_t st, const FsGrid& g, const float* p, float* np, const float* div,
                  int zb, int ze);
// `sweeps` Jacobi sweeps with temporal blocking where it pays; returns through *result_in_np
// whether the final iterate ended in np (1) or p (0), and through *prev_held whether the other
// buffer holds the iterate before the last sweep (true iff the last launch was a single sweep).
// Single-slab (no halo exchange) only.
void jacobi_multi(hipStream_t st, const FsGrid& g, float* p, float* np, const float* div,
                  int sweeps, int naive, int* result_in_np, int* prev_held);
// One pass of `sweeps` (1..jacobi_block()) fused sweeps p -> np: level-0 planes [zin_lo, zin_hi)
// are read (ghost planes included), planes [zout_lo, zout_hi) are written.  A plane range edge
// that is not a global wall loses one plane of validity per sweep, so zout must lie `sweeps`
// planes inside zin there.  [zout2_lo, zout2_hi) is an optional second range produced by the same
// launch (the two boundary layers of a slab).  Returns false if this grid cannot take the fused
// kernel.
bool jacobi_pass(hipStream_t st, const FsGrid& g, const float* p, float* np, const float* div,
                 int sweeps, int zin_lo, int zin_hi, int zout_lo, int zout_hi, int zout2_lo = 0,
                 int zout2_hi = 0);
bool jacobi_can_block(const FsGrid& g);
// The same with the six-sweep kernel: one pass of jacobi_deep_block() sweeps, zout that many planes
// inside zin wherever the edge is not a global wall.  False if this grid cannot take the kernel.
bool jacobi_deep_pass(hipStream_t st, const FsGrid& g, const float* p, float* np, const float* div,
                      int zin_lo, int zin_hi, int zout_lo, int zout_hi, int zout2_lo = 0, int zout2_hi = 0);
int jacobi_deep_block();   // sweeps fused per six-sweep pass
// Whether a z-slab of this decomposition runs six-sweep passes.  Every rank must make the same
// sequence of exchanges, so this is decided by what all ranks see alike: the grid, the thinnest slab
// and the environment, never by a rank's own timing.
bool jacobi_deep_slab(const FsGrid& g);
// Six-sweep passes among `sweeps` sweeps when the rest runs in passes of `unit` sweeps: as many as
// leave a multiple of `unit`, if giving up at most one achieves that; else as many as fit.
int jacobi_deep_count(int sweeps, int unit);
// Times the tile variants of the blocked kernel on this grid (p, np, div must hold zeros: the
// passes then leave them zero) and their z chunking, and records the fastest in g.jacobi_tile /
// g.jacobi_lz.  All variants return the same values, so the choice only affects speed.
void jacobi_autotune(hipStream_t st, FsGrid& g, float* p, float* np, const float* div);
// what the blocked passes of this grid run with: tile variant (-1 = the grid cannot block), its
// tile rows, whether its output rows are 64-byte aligned, planes per z chunk (0 = cost model)
void jacobi_config(const FsGrid& g, int* variant, int* rows, int* aligned, int* planes_per_chunk);
int jacobi_nontemporal(const FsGrid& g);
// the deepest pass of this grid: sweeps it fuses (6 with the six-sweep passes on, else 4, or 1 if the
// grid cannot block), its planes per z chunk and store policy when that is the six-sweep pass
void jacobi_deep_config(const FsGrid& g, int* sweeps_per_pass, int* planes_per_chunk, int* nontemporal);
// six-sweep passes among the launches jacobi_multi makes for `sweeps` sweeps
int jacobi_deep_passes(const FsGrid& g, int sweeps, int naive);
int jacobi_block();   // sweeps fused per pass
// number of kernel launches jacobi_multi makes for `sweeps` sweeps
int jacobi_launches(const FsGrid& g, int sweeps, int naive);
void gradient(hipStream_t st, const FsGrid& g, float* ux, float* uy, float* uz, const float* p);
// Multigrid-preconditioned CG (single slab only); dense work arrays, see fs_mgpcg.hip.
size_t mg_space(const FsGrid& g, int levels);       // floats in r and in z (all levels)
int cg_dot_partials();                               // doubles the dot-product scratch must hold
void cg_init(hipStream_t st, const FsGrid& g, float* r, float* x, const float* div);
void cg_to_pressure(hipStream_t st, const FsGrid& g, float* p, const float* x);
// A CG coefficient (alpha, beta): either a value the host has, or the quotient of two dot products
// that are still on the device -- each narrowed to binary32 first, then divided in binary32, which is
// what the reference's host code does with the numbers it has read back (Solver.cu:1263, 1288).
struct CgScalar {
    float v;
    const double *num, *den;
    CgScalar(float value) : v(value), num(nullptr), den(nullptr) {}
    CgScalar(const double* n, const double* d) : v(0.f), num(n), den(d) {}
};
void cg_apply_a(hipStream_t st, const FsGrid& g, float* Ap, const float* p);
void cg_update_xr(hipStream_t st, const FsGrid& g, float* x, float* r, const float* p, const float* Ap, CgScalar alpha);
void cg_update_p(hipStream_t st, const FsGrid& g, float* p, const float* z, CgScalar beta);
void cg_dot(hipStream_t st, const FsGrid& g, const float* a, const float* b, double* partial, double* out);
// (k_lo, k_hi: the planes of g whose products enter the sum -- a slab's own planes inside its window;
//  the operator and the update cover all of g either way; k_hi < 0 = all planes)
void cg_apply_a_dot(hipStream_t st, const FsGrid& g, float* Ap, const float* p, double* partial, double* out,
                    int k_lo = 0, int k_hi = -1);
void cg_update_p_apply_a_dot(hipStream_t st, const FsGrid& g, float* Ap, float* pn, const float* po, const float* z,
                             CgScalar beta, double* partial, double* out);
void cg_update_xr_dot(hipStream_t st, const FsGrid& g, float* x, float* r, const float* p, const float* Ap,
                      CgScalar alpha, double* partial, double* out, int k_lo = 0, int k_hi = -1);
float* mg_precondition(hipStream_t st, const FsGrid& g, float* r, float* z, float* zalt, int levels,
                       int init_smooth_steps, int bottom_smooth_steps);   // returns z or zalt: the result
// The same operations on a z-slab's windows of the levels (dense nx x ny x held arrays that hold the
// owned planes plus ghost planes; see fs_mgpcg.hip): one colour of the smoother (kpar = parity of the
// global index of the window's plane 0), restriction into the cz owned coarse planes (fk0 = fine
// window plane of the lower children of the first of them), prolongation over every fine plane whose
// parent the coarse window holds (f0 / c0 = global index of the fine / coarse window's plane 0).
void mgs_smooth_phase(hipStream_t st, const float* r, float* z, int nx, int ny, int held, int phase, int kpar);
bool mgs_smooth_pass(hipStream_t st, const FsEnv& env, const float* r, const float* z, float* zalt, int nx, int ny,
                     int held, int colours, bool z_is_zero, int kpar);
void mgs_restrict(hipStream_t st, const float* rf, const float* zf, int nx, int ny, int held_f, int fk0,
                  float* rc_owned, int cz);
void mgs_prolongate(hipStream_t st, float* zf, int nx, int ny, int held_f, int f0, const float* zc, int c0, int cheld);
// `phases` (2 or 4) red/black Gauss-Seidel half-sweeps of the zero-outside operator on a dense
// nx x ny x nz array in one blocked pass, zin -> zout; false if the grid cannot take it
// (zin_is_zero: zin holds zeros and is not read; kpar: parity of the global z index of plane 0, which
//  enters the colour of a cell -- a slab's window need not start on an even plane)
bool rbgs_pass(hipStream_t st, const FsEnv& env, int nx, int ny, int nz, const float* rhs, const float* zin, float* zout,
               int phases, bool zin_is_zero = false, int kpar = 0);
// Vorticity confinement.  The three vorticity arrays are dense, vort_elems(g) floats each, zero-filled
// once by the owner; on a slab they are this rank's window of the global flat index space: its
// cells plus vort_halo_planes(g) planes of nx*ny entries per side, which the neighbours fill.
size_t vort_elems(const FsGrid& g);
int vort_halo_planes(const FsGrid& g);
void find_vort(hipStream_t st, const FsGrid& g, float* vortx, float* vorty, float* vortz, const float* ux,
               const float* uy, const float* uz);
// lattice: vort_lattice_elems(g) floats of scratch for the |omega| values the faces share; without
// it every face evaluates its own (one launch per component)
size_t vort_lattice_elems(const FsGrid& g);
void vort_confine(hipStream_t st, const FsGrid& g, float* ux, float* uy, float* uz, const float* vortx,
                  const float* vorty, const float* vortz, float coeff, float* lattice = nullptr);
// sum over owned cells of (div + 6p - neighbours)^2 in binary64; partial must hold
// residual_partials(g) doubles, out one double.
int residual_partials(const FsGrid& g);
void residual(hipStream_t st, const FsGrid& g, const float* p, const float* div, double* partial,
              double* out);
// The same scratch as a file-scope device array, for callers that must not allocate: one copy per
// GPU, residual_static_rows() partials plus the total.  residual_static_lookup(device) resolves the
// array's address on `device`, which must be the current device (once per device; later calls
// return at once); residual_static_scratch hands out the resolved addresses, false if the lookup
// has not run for that device.
int residual_static_rows();
hipError_t residual_static_lookup(int device);
bool residual_static_scratch(int device, double** partial, double** out);

}  // namespace fsk
