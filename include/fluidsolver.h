/*
 * fluidsolver.h -- C ABI of the MI355X (gfx950) smoke-solver step.
 *
 * This is the drop-in boundary under the reference's C++ `Solver` class
 * (reference: CuSolver/Include/Solver.h:13-167).  Plain pointers and sizes only;
 * every function returns 0 on success or a negative fs_status, and stores a
 * message retrievable with fs_last_error() (thread-local).
 *
 * Threading contract (same as the reference: one host thread drives one solver,
 * CuSolver/Source/Solver.cu uses the default stream only): calls on one
 * fs_solver must not overlap.  fs_step() only enqueues GPU work; the *_host
 * getters, fs_sync() and fs_residual() synchronise.
 *
 * All fields are fp32 on a staggered grid, h = 1 (Solver.cu:17-27, 377-390):
 *   FS_UX (nx+1,ny,nz)  FS_UY (nx,ny+1,nz)  FS_UZ (nx,ny,nz+1)
 *   FS_RHO, FS_T, FS_P, FS_DIV (nx,ny,nz)
 * Host-side buffers use the reference's dense layout, flat index
 * x + y*dx + z*dx*dy (Solver.cu:33).  Device buffers are padded; see fs_layout.
 */
#ifndef FLUIDSOLVER_H
#define FLUIDSOLVER_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(_WIN32)
#define FS_API
#else
#define FS_API __attribute__((visibility("default")))
#endif

typedef enum fs_status {
    FS_OK = 0,
    FS_ERR_INVALID = -1,     /* bad argument */
    FS_ERR_HIP = -2,         /* a HIP runtime call failed */
    FS_ERR_NO_DEVICE = -3,   /* no usable gfx950 device */
    FS_ERR_HALO = -4,        /* a back-trace left the ghost planes of a z-slab */
    FS_ERR_COMM = -5,        /* RCCL failure */
    FS_ERR_STATE = -6        /* call not valid in this state */
} fs_status;

/* field selectors */
enum { FS_UX = 0, FS_UY = 1, FS_UZ = 2, FS_RHO = 3, FS_T = 4, FS_P = 5, FS_DIV = 6, FS_NUM_FIELDS = 7 };

/* pressure solver (Solver.cu:1178-1192): the Jacobi branch, or the shipped multigrid-
   preconditioned conjugate gradient (Conjugate :1226-1297, MG_Preconditioner :1312-1366) */
enum { FS_PRESSURE_JACOBI = 0, FS_PRESSURE_MGPCG = 1 };

/* back-trace scheme of the semi-Lagrangian advection (Solver.cu:11, 393-397) */
enum { FS_SCHEME_RK4 = 0, FS_SCHEME_MACCORMACK = 1 };

/* fs_params.flags */
enum {
    FS_FLAG_NO_GRAPH = 1 << 0,   /* launch kernels eagerly instead of replaying a hipGraph */
    FS_FLAG_JACOBI_NAIVE = 1 << 1, /* one kernel launch per sweep, no temporal blocking */
    /* Vorticity confinement, Solver.cu:13 VORT_CONFINE (shipped as 1; off on the Jacobi hot path):
       FindVortx/y/z after the buoyancy (Solver.cu:1035-1037), ApplyVortConfinement after the
       advection (Solver.cu:1078-1080).  The reference gathers from its vorticity arrays beyond
       their allocation (Solver.cu:556-558 against :894-896); this build gives the same index
       arithmetic arrays of (nx+1)(ny+1)(nz+1) zero-initialised floats, so every read is defined.
       On z-slabs every rank holds its window of those arrays plus a halo from its neighbours
       (the mismatched strides make a gather land up to about 2 + (nx + ny) nz / (nx ny) planes
       above its face; fs_create_slab refuses decompositions whose slabs are thinner than that). */
    FS_FLAG_VORT_CONFINE = 1 << 2,
    /* Test hook of z-slabs: the slab keeps its place in the decomposition but both neighbours are
       the rank itself (RCCL transport: its communicator holds this rank alone; host-staged
       transport: the exchange function is expected to hand send_hi back as recv_lo and send_lo as
       recv_hi), so one process on one GPU can drive the transfers and the stream ordering.  The
       results are NOT those of the decomposed grid; never set it in a real multi-rank run. */
    FS_FLAG_TEST_SELF_PEER = 1 << 3,
    /* The arithmetic of a contracting build of the reference.  Its binary is built by nvcc, whose
       default -fmad=true fuses a multiply into a following add; this library's default is the
       uncontracted meaning of the source text (every product and sum rounded on its own).  With
       this flag the expressions of the advection, the buoyancy and the confinement gathers that
       have the shape x*y + z are evaluated with one fused multiply-add, the way an LLVM-based
       compiler contracts them: lerp (Math.cuh:57-60) as fma(a, 1-s, b*s), the RK position updates
       (Solver.cu:145, 150, 155) as fma(-c, u, pos), the buoyancy update (Solver.cu:360-361) as
       fma(T - T_env, dt, uy).  Which products the reference's own binary fuses is not determined by
       its source; this is one defined member of that family (the two meanings differ by about 1e-4
       of the field maximum in the density and 1e-3 in the velocity after 100 steps at 64^3). */
    FS_FLAG_FMAD = 1 << 4
};

/*
 * Run-time form of what the reference fixes at compile time:
 * grid size macros Nx,Ny,Nz (Solver.h:9-11) and the member initialisers
 * Solver.h:58-81.  fs_default_params() fills the reference values.
 */
typedef struct fs_params {
    int nx, ny, nz;              /* Solver.h:58-60; 1 <= n <= 4096 each */
    float dt;                    /* Solver.h:71   0.2  */
    int jacobi_iters;            /* Solver.h:68   max_iter = 30, sweeps per step */
    int scheme;                  /* Solver.cu:11  FS_SCHEME_MACCORMACK */
    /* Source sphere (SourceKernel, Solver.cu:295-307).  The test (x-sx)^2 + (y-sy)^2 + (z-sz)^2 <= r^2
       squares binary32 differences; the reference writes pow(v, 2).  The two agree whenever the
       differences square exactly in binary32, which holds for centres on multiples of 1/2 (the
       reference's own Nx/2, 10, Nz/2) and fractions with a few bits; for other centres a cell on
       the rim of the sphere may fall on the other side. */
    float source_x;              /* Solver.h:78   Nx/2 */
    float source_y;              /* Solver.h:79   10   */
    float source_z;              /* Solver.h:80   Nz/2 */
    float source_radius;         /* Solver.h:81   7    */
    float source_rho;            /* Solver.h:62   0.1  */
    float source_temperature;    /* Solver.h:64   2.0  */
    float temperature_env;       /* Solver.h:65   0.0  */
    int device;                  /* HIP device ordinal; -1 = current device */
    unsigned flags;              /* FS_FLAG_* */
    /* convergence mode (weak-scaling config): if tolerance > 0, each step sweeps
       in groups of check_every until sum r^2 < tolerance or max_sweeps is hit.
       jacobi_iters is ignored in that mode. */
    float tolerance;
    int check_every;
    int max_sweeps;
    float curl_strength;         /* Solver.h:73   0.1, used with FS_FLAG_VORT_CONFINE */
    /* FS_PRESSURE_MGPCG (every dim a multiple of 2^(mg_level-1); on z-slabs nz a multiple of
       ranks x 2^(mg_level-1), so that every slab coarsens on its own, and the dot products are sums
       of the ranks' partial sums): jacobi_iters is the
       iteration limit, as max_iter is for both branches in the reference (Solver.cu:1256).  The
       solve follows the reference literally, including that the first solve after creation does
       not reach the pressure field (the swap at Solver.cu:1295 exchanges a by-value pointer). */
    int pressure_solver;         /* FS_PRESSURE_*  default FS_PRESSURE_JACOBI */
    int mg_level;                /* Solver.h:151  4  */
    int init_smooth_steps;       /* Solver.h:153  2  */
    int bottom_smooth_steps;     /* Solver.h:154  50 */
    float iter_precision;        /* Solver.h:69   1e-3, stop when r.r falls below */
} fs_params;

/* Device-side layout of one field (all fields of a solver share the strides). */
typedef struct fs_layout {
    int dx, dy, dz;              /* logical dims of the field (global) */
    int z0;                      /* global z index of the first plane held locally */
    int nz_local;                /* number of owned planes of this field on this rank */
    int ghost;                   /* readable ghost planes on each z side */
    size_t pitch;                /* floats between consecutive y rows     */
    size_t plane;                /* floats between consecutive z planes   */
} fs_layout;

typedef struct fs_solver fs_solver;

/* -- lifetime: replaces Solver::Solver/Initialize/~Solver (Solver.cu:1116-1143) -- */
FS_API int fs_default_params(fs_params* p, int nx, int ny, int nz);
/* Allocates and zeroes every device field, sets T = temperature_env
   (Solver.cu:879-952, 993-995). */
FS_API int fs_create(const fs_params* p, fs_solver** out);
FS_API int fs_destroy(fs_solver* s);

/* -- z-slab decomposition over the GPUs of one node, one process per GPU (no counterpart in
      the reference, which is single-GPU).  Rank r owns cell planes [z0, z0 + nz_local) of the
      global grid p->nz is cut into; every other call below then acts on that slab and the
      host getters / fs_field_upload move the LOCAL planes only (nz_local planes, nz_local + 1
      for FS_UZ).  Results are bit-identical to the single-GPU solver (with FS_PRESSURE_MGPCG: up
      to the summation order of the dot products, which are reduced over the ranks).
      Transport, one of:
        * unique_id != NULL: RCCL.  Rank 0 obtains 128 bytes from fs_comm_unique_id and hands
          them to every rank (any out-of-band channel); halos travel as ncclSend/ncclRecv
          between z-neighbours, the residual norm and MGPCG's dot products as ncclAllReduce.
        * unique_id == NULL, fn != NULL: host-staged.  The library packs halos into pinned
          host buffers and calls fn to move them (MPI, gloo, ...).  Used by the tests and as a
          fallback; fn(ctx, FS_XCHG_HALO, send_lo, recv_lo, send_hi, recv_hi, bytes) must
          deliver send_lo to rank-1 / send_hi to rank+1 and fill recv_* from them (a NULL pair
          means no neighbour on that side); fn(ctx, FS_XCHG_SUM_F64, value, 0, 0, 0, 8) must
          replace *(double*)value by its sum over all ranks.  Return 0 on success. -- */
#define FS_UNIQUE_ID_BYTES 128
enum { FS_XCHG_HALO = 0, FS_XCHG_SUM_F64 = 1 };
typedef int (*fs_exchange_fn)(void* ctx, int op, void* send_lo, void* recv_lo, void* send_hi,
                              void* recv_hi, size_t bytes);
FS_API int fs_comm_unique_id(void* id /* FS_UNIQUE_ID_BYTES */);
FS_API int fs_create_slab(const fs_params* p, int rank, int world, const void* unique_id,
                          fs_exchange_fn fn, void* ctx, fs_solver** out);
FS_API int fs_slab_info(const fs_solver* s, int* rank, int* world, int* z0, int* nz_local);
/* ranks of the RCCL communicator the slab exchanges over (ncclCommCount); 0 = no communicator
   (single GPU or host-staged transport), -1 = error */
FS_API int fs_comm_ranks(const fs_solver* s);
/* What the temporally blocked Jacobi passes of this solver run with (chosen by timing at creation,
   no counterpart in the reference): tile variant 0..3 (-1 = this grid takes single sweeps only),
   rows per tile, whether output rows are 64-byte aligned, planes per z chunk (0 = cost model),
   whether the passes store with the non-temporal cache policy.  Any pointer may be NULL. */
FS_API int fs_jacobi_config(const fs_solver* s, int* variant, int* tile_rows, int* aligned_rows,
                            int* planes_per_chunk, int* nontemporal_stores);
/* The deepest pass this solver's Jacobi sweeps run in (no counterpart in the reference): sweeps
   fused per pass (6 where the timing at creation found the six-sweep passes faster on a single
   domain, else 4, 1 on grids the blocked kernels cannot take) and, for the six-sweep pass, its
   planes per z chunk (0 = cost model) and store policy.  Any pointer may be NULL. */
FS_API int fs_jacobi_deep_config(const fs_solver* s, int* sweeps_per_pass, int* planes_per_chunk,
                                 int* nontemporal_stores);

/* -- stepping: replaces Solver::Update -> UpdateCuda (Solver.cu:1009-1114, 1132-1137)
      with VORT_CONFINE off and Project() on its Jacobi branch (Solver.cu:1178-1189) -- */
FS_API int fs_step(fs_solver* s);
FS_API int fs_steps(fs_solver* s, int n);
FS_API int fs_sync(fs_solver* s);
FS_API int fs_frame(const fs_solver* s);          /* Solver.h:76 current_frame */
/* fs_step replays a captured hipGraph per assignment of the buffers where it can (one GPU, fixed
   sweep count, no stage timing; z-slabs launch eagerly).
   Returns the number of step graphs this solver holds (no counterpart in the reference, which
   launches eagerly). */
FS_API int fs_step_graphs(const fs_solver* s);

/* -- individual stages, in the order fs_step runs them.  The reference exposes
      Advect() and Project() as public members (Solver.h:24,26); the others are its
      kernel launches at Solver.cu:1028, 1031 and 1111-1113. -- */
FS_API int fs_stage_source(fs_solver* s);         /* SourceKernel      Solver.cu:295-307 */
FS_API int fs_stage_force(fs_solver* s);          /* ForceKernelUy     Solver.cu:346-368 */
FS_API int fs_stage_vorticity(fs_solver* s);      /* FindVortx/y/z     Solver.cu:485-525, 1035-1037; no-op without FS_FLAG_VORT_CONFINE */
FS_API int fs_stage_advect(fs_solver* s);         /* Solver::Advect    Solver.cu:1145-1165 */
FS_API int fs_stage_confine(fs_solver* s);        /* ApplyVortConfinement Solver.cu:540-605, 1078-1080; no-op without the flag */
FS_API int fs_stage_divergence(fs_solver* s);     /* DivergenceKernel  Solver.cu:406-423 */
FS_API int fs_stage_jacobi(fs_solver* s, int sweeps); /* JacobiKernel + swap, Solver.cu:426-443, 1180-1184 */
FS_API int fs_stage_project(fs_solver* s);        /* Solver::Project   Solver.cu:1168-1189 */
/* The solve of Project() alone, on the divergence field as it stands: the Jacobi loop
   (Solver.cu:1180-1184) or Solver::Conjugate(f_pressure, f_div) (Solver.cu:1226-1297, public in
   Solver.h:37), by fs_params.pressure_solver. */
FS_API int fs_stage_pressure_solve(fs_solver* s);
/* Solver::MG_Preconditioner (Solver.cu:1312-1366, public in Solver.h:40): one V-cycle z = M^-1 r on
   the solver's own multigrid arrays.  FS_PRESSURE_MGPCG solvers only, else FS_ERR_STATE. */
FS_API int fs_mg_vcycle(fs_solver* s);
FS_API int fs_stage_gradient(fs_solver* s);       /* ApplyGradientU*   Solver.cu:446-483 */

/* -- buffer access: replaces Solver::GetDensityField (Solver.cu:1368-1372).
      Returned host pointers are solver-owned (pinned), dense reference layout,
      valid until the next getter call for the same field or fs_destroy. -- */
FS_API const float* fs_density_host(fs_solver* s);
FS_API const float* fs_velocity_host(fs_solver* s, int component /* 0,1,2 */);
FS_API const float* fs_field_host(fs_solver* s, int field);
FS_API size_t fs_field_size(const fs_solver* s, int field);      /* element count, dense */
/* Asynchronous form of fs_field_host (no counterpart in the reference, whose blocking copy at
   Solver.cu:1370 stalls the render loop): _begin takes a device-side snapshot of the field in
   stream order and starts its copy to the pinned host buffer on a separate stream, so fs_step
   calls issued afterwards overlap with the transfer; _end waits for it and returns the buffer.
   One read-back in flight at a time. */
FS_API int fs_field_host_begin(fs_solver* s, int field);
FS_API const float* fs_field_host_end(fs_solver* s, int field);
/* Upload a dense host array into a field (tests, restart from a dump). */
FS_API int fs_field_upload(fs_solver* s, int field, const float* src);
/* Set every element of a field (this rank's planes) to one value: what Solver::ResetGridValue
   (Solver.cu:1204-1210, kernels :209-229) does with 0 for the density and the velocity. */
FS_API int fs_field_fill(fs_solver* s, int field, float value);
/* Zero-copy access for device-side consumers. *ptr addresses element (0,0,z0).  The pointer is
   valid until the next call that steps or uploads (buffers are swapped); a consumer that writes
   through it must call this function again before the next step so that ghost planes and
   derived fields are refreshed. */
FS_API int fs_field_device(fs_solver* s, int field, float** ptr, fs_layout* layout);

/* -- residual: what Mul + Reduce are meant to yield at Solver.cu:1185-1187:
      sum over the grid of r^2 with r = div + 6p - sum(neighbours) (Solver.cu:442),
      accumulated in binary64.  which = 0: r of the current pressure;
      which = 1: r of the input iterate of the last sweep (what the reference's r
      array holds after its loop); available only while that iterate is still held,
      i.e. when the last sweep ran on its own (sweep count not a multiple of 4, or
      FS_FLAG_JACOBI_NAIVE), else FS_ERR_STATE. -- */
FS_API int fs_residual(fs_solver* s, int which, double* out);
/* fs_residual for callers that must not allocate device or pinned memory (tight serving loops,
   harnesses that forbid hipMalloc; the call still synchronises its stream like fs_residual, so
   while another thread captures a HIP graph that thread must capture in thread-local or relaxed
   mode).  No counterpart in the reference.  Same arguments, result and synchronisation as fs_residual, but the partial
   sums live in a file-scope device array of the library (FS_NOALLOC_MAX_ROWS partials + the total,
   8 MiB per GPU, held until the process ends) instead of the solver's own scratch buffer, and the
   call makes no hipMalloc / hipFree / pinned allocation.  Limits: single-GPU solvers (not z-slabs:
   FS_ERR_STATE) with ny * nz <= FS_NOALLOC_MAX_ROWS (else FS_ERR_INVALID).  All solvers on one GPU
   share that GPU's array: calls that overlap in time on one GPU race; calls on one solver, or on
   several solvers from one thread, are ordered by the synchronisation at the end of each call. */
#define FS_NOALLOC_MAX_ROWS (1 << 20)
FS_API int fs_residual_noalloc(fs_solver* s, int which, double* out);
/* sweeps (Jacobi) or CG iterations (MGPCG) the last fs_step / fs_stage_project actually ran */
FS_API int fs_last_sweeps(const fs_solver* s);
/* r.r after the last MGPCG solve (what the reference prints as rTr, Solver.cu:1272) */
FS_API int fs_cg_residual(const fs_solver* s, float* rtr);

/* -- stage timing (no counterpart in the reference, which only prints fps,
      Renderer/Source/FluidSolver.cpp:179): when enabled, every stage is bracketed by a
      pair of HIP events on the solver's own stream.  fs_profile_read synchronises and
      returns the accumulated device time of a stage and the number of kernel launches
      it made, so  ms_total / launches  is the average launch duration. -- */
enum {
    FS_STAGE_SOURCE = 0, FS_STAGE_FORCE, FS_STAGE_ADVECT_VELOCITY, FS_STAGE_ADVECT_SCALARS,
    FS_STAGE_DIVERGENCE, FS_STAGE_JACOBI, FS_STAGE_GRADIENT, FS_STAGE_VORTICITY, FS_STAGE_CONFINE,
    FS_STAGE_MGPCG, FS_NUM_STAGES
};
FS_API int fs_profile_enable(fs_solver* s, int on);
FS_API int fs_profile_read(fs_solver* s, int stage, double* ms_total, long long* launches);
FS_API int fs_profile_reset(fs_solver* s);
/* How long a z-slab waited for halos since the last fs_profile_reset (no counterpart in the
   reference, which is single-GPU): host_ms = host time inside exchange calls that block (the
   host-staged transport's pack / callback, a host-side wait for received planes); device_ms = time
   the compute stream stood still between asking for received planes and having them (collected
   only while stage timing is enabled; synchronises); exchanges = exchanges begun.  Any pointer may
   be NULL.  All zero on a single-GPU solver. */
FS_API int fs_halo_wait(fs_solver* s, double* host_ms, double* device_ms, long long* exchanges);

FS_API const char* fs_last_error(void);
FS_API const char* fs_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FLUIDSOLVER_H */
