// fs_solver.hip -- the solver object behind the C ABI of include/fluidsolver.h.
//
// Host-side orchestration of one simulation step on one z-slab: buffer ownership,
// stage order (reference: CuSolver/Source/Solver.cu:1009-1114 UpdateCuda, 1145-1165
// Advect, 1168-1189 Project on its Jacobi branch), ping-pong swaps (Solver.cu:181-186),
// host getters (Solver.cu:1368-1372).
//
// The other translation units behind the C ABI: fs_exchange.hip (halo exchange of a z-slab),
// fs_project_jacobi.hip / fs_project_mgpcg.hip (the pressure solves), fs_stages.hip (stage entry
// points), fs_step.hip (a step, eager or as a graph replay); fs_solver_impl.h is what they share.
#include "fs_solver_impl.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <utility>

using namespace fsi;

namespace fsi {

std::string& error_text()
{
    thread_local std::string text;
    return text;
}

int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    error_text() = buf;
    return code;
}

int check_flag(fs_solver* s)
{
    // only slabs with a non-wall z face can raise the halo flag
    if (s->g.wall_lo && s->g.wall_hi) return FS_OK;
    FS_HIP(hipMemcpyAsync(s->h_flag, s->d_flag, sizeof(int), hipMemcpyDeviceToHost, s->st));
    FS_HIP(hipStreamSynchronize(s->st));
    if (*s->h_flag)
        return fail(FS_ERR_HALO, "rank %d: a back-trace left the %d ghost planes of the z-slab",
                    s->rank, s->g.ghost);
    return FS_OK;
}

}  // namespace fsi

namespace {

// dims of the LOCAL part of a field (what the host getters return)
void field_dims(const fs_solver* s, int field, int& dx, int& dy, int& dz)
{
    dx = s->g.nx + (field == FS_UX);
    dy = s->g.ny + (field == FS_UY);
    dz = s->g.nzl + (field == FS_UZ);
}

float** field_slot(fs_solver* s, int field)
{
    switch (field) {
    case FS_UX: return &s->f.ux;
    case FS_UY: return &s->f.uy;
    case FS_UZ: return &s->f.uz;
    case FS_RHO: return &s->f.rho;
    case FS_T: return &s->f.T;
    case FS_P: return &s->f.p;
    case FS_DIV: return &s->f.div;
    default: return nullptr;
    }
}

int create_common(const fs_params* p, int rank, int world, const void* unique_id, fs_exchange_fn fn,
                  void* ctx, fs_solver** out)
{
    if (!p || !out) return fail(FS_ERR_INVALID, "fs_create: null argument");
    *out = nullptr;
    if (p->nx < 1 || p->ny < 1 || p->nz < 1 || p->nx > 4096 || p->ny > 4096 || p->nz > 4096)
        return fail(FS_ERR_INVALID, "fs_create: grid %dx%dx%d outside 1..4096", p->nx, p->ny, p->nz);
    if (p->jacobi_iters < 0) return fail(FS_ERR_INVALID, "fs_create: jacobi_iters < 0");
    if (p->scheme != FS_SCHEME_RK4 && p->scheme != FS_SCHEME_MACCORMACK)
        return fail(FS_ERR_INVALID, "fs_create: unknown scheme %d", p->scheme);
    if (p->tolerance > 0.f && (p->check_every < 1 || p->max_sweeps < 1))
        return fail(FS_ERR_INVALID, "fs_create: convergence mode needs check_every, max_sweeps >= 1");
    if (world < 1 || rank < 0 || rank >= world)
        return fail(FS_ERR_INVALID, "fs_create_slab: rank %d of %d", rank, world);
    if (world > 1 && !unique_id && !fn)
        return fail(FS_ERR_INVALID, "fs_create_slab: neither an RCCL id nor an exchange function");
    if (p->pressure_solver != FS_PRESSURE_JACOBI && p->pressure_solver != FS_PRESSURE_MGPCG)
        return fail(FS_ERR_INVALID, "fs_create: unknown pressure_solver %d", p->pressure_solver);
    if (p->pressure_solver == FS_PRESSURE_MGPCG) {
        if (p->mg_level < 1 || p->mg_level > 8 || p->init_smooth_steps < 0 || p->bottom_smooth_steps < 0)
            return fail(FS_ERR_INVALID, "fs_create: bad multigrid parameters");
        if (world > 1 && p->nz % (world << (p->mg_level - 1)) != 0)
            return fail(FS_ERR_INVALID, "fs_create_slab: FS_PRESSURE_MGPCG needs nz to be a multiple of %d "
                                        "(ranks x 2^(mg_level-1): every slab must coarsen on its own)",
                        world << (p->mg_level - 1));
        const int m = 1 << (p->mg_level - 1);
        if (p->nx % m || p->ny % m || p->nz % m)
            return fail(FS_ERR_INVALID, "fs_create: FS_PRESSURE_MGPCG needs every dim to be a multiple of %d "
                                        "(mg_level %d)", m, p->mg_level);
        if (p->tolerance > 0.f) return fail(FS_ERR_INVALID, "fs_create: convergence mode is for the Jacobi solver");
    }
    if ((p->flags & FS_FLAG_TEST_SELF_PEER) && world < 2)
        return fail(FS_ERR_INVALID, "fs_create: FS_FLAG_TEST_SELF_PEER is a test hook of z-slabs");
    if (world > 1 && (p->flags & FS_FLAG_VORT_CONFINE)) {
        // the confinement gathers land up to this many xy-planes above a rank's own entries of the
        // vorticity arrays (mismatched strides, see fs_vort.hip); they must come from the direct
        // neighbour
        FsGrid probe{};
        probe.nx = p->nx; probe.ny = p->ny; probe.nz = p->nz;
        const int h = fsk::vort_halo_planes(probe);
        if (h > p->nz / world)
            return fail(FS_ERR_INVALID, "fs_create_slab: FS_FLAG_VORT_CONFINE on %dx%dx%d needs %d planes of the "
                                        "upper neighbour's vorticity, slabs of %d ranks hold only %d",
                        p->nx, p->ny, p->nz, h, world, p->nz / world);
    }
    // planes an advection gather may reach into a neighbour's slab / ghost planes held per side.
    // A slab holds twice the depth of a fused Jacobi pass where it is thick enough: two passes then
    // share one exchange (FS_SLAB_PGHOST=4 keeps one exchange per pass).
    // the environment switches are read here, once per solver, and travel inside the grid
    const FsEnv env = fsk::read_env();
    const int ghost = world > 1 ? 4 : 2;
    int pghost = ghost;
    if (world > 1 && env.slab_pghost >= 8 && p->nz / world >= 9) pghost = 8;
    // z-slabs: the first nz % world ranks get one plane more
    const int base = p->nz / world, extra = p->nz % world;
    const int nzl = base + (rank < extra ? 1 : 0);
    const int z0 = rank * base + (rank < extra ? rank : extra);
    if (world > 1 && base < ghost + 1)
        return fail(FS_ERR_INVALID, "fs_create_slab: %d planes over %d ranks leaves slabs thinner than "
                                    "the %d ghost planes + 1", p->nz, world, ghost);

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(FS_ERR_NO_DEVICE, "fs_create: no HIP device visible");
    int dev = p->device;
    if (dev < 0) FS_HIP(hipGetDevice(&dev));
    if (dev >= ndev) return fail(FS_ERR_NO_DEVICE, "fs_create: device %d of %d", dev, ndev);
    FS_HIP(hipSetDevice(dev));
    hipDeviceProp_t prop;
    FS_HIP(hipGetDeviceProperties(&prop, dev));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(FS_ERR_NO_DEVICE, "fs_create: device %d is %s; this library holds gfx950 code only",
                    dev, prop.gcnArchName);

    // address of the static residual scratch on this device (fs_residual_noalloc), looked up here
    // so that the entry point itself never has to.  A failed lookup (e.g. a device ordinal beyond
    // the table) only leaves that one entry point without its scratch, which it reports itself.
    (void)fsk::residual_static_lookup(dev);

    fs_solver* s = new (std::nothrow) fs_solver;
    if (!s) return fail(FS_ERR_INVALID, "fs_create: out of host memory");
    s->prm = *p;
    s->device = dev;
    s->rank = rank;
    s->world = world;
    s->xfn = fn;
    s->xctx = ctx;
    FsGrid& g = s->g;
    g.env = env;
    g.fmad = (p->flags & FS_FLAG_FMAD) ? 1 : 0;
    g.nx = p->nx;
    g.ny = p->ny;
    g.nz = p->nz;
    g.z0 = z0;
    g.nzl = nzl;
    g.nz_min = base;
    g.ghost = ghost;
    g.pghost = pghost;
    g.wall_lo = rank == 0;
    g.wall_hi = rank == world - 1;
    // an odd multiple of 16 floats: consecutive rows start in alternate halves of a 128-byte
    // line, which measures ~8 % faster on the stencil kernels than line-aligned rows
    g.pitch = round_up(p->nx + 3, 16);
    if (g.pitch % 32 == 0) g.pitch += 16;
    long long plane = (long long)g.pitch * (p->ny + 3);
    long long planes = (long long)g.nzl + 1 + 2 * g.pghost;
    // kernels index with 32-bit element offsets relative to global plane 0
    if (plane * ((long long)p->nz + g.pghost + 2) >= (1LL << 31)) {
        delete s;
        return fail(FS_ERR_INVALID, "fs_create: %dx%dx%d needs more than 2^31 elements per field",
                    p->nx, p->ny, p->nz);
    }
    g.plane = (int)plane;
    // slack so that (x,y,z) = (-2,-2,z0-ghost) is addressable, kept 256-byte aligned
    size_t slack = (size_t)round_up(2 * g.pitch + 2, 64);
    s->origin = slack + (size_t)g.pghost * plane;
    s->field_elems = (slack + (size_t)(planes * plane) + 63) / 64 * 64;

    const int nfields = 13;
    int rc = FS_OK;
    do {
        if (hipStreamCreateWithFlags(&s->st, hipStreamNonBlocking) != hipSuccess) {
            rc = fail(FS_ERR_HIP, "fs_create: hipStreamCreate failed");
            break;
        }
        size_t bytes = s->field_elems * nfields * sizeof(float);
        hipError_t e = hipMalloc((void**)&s->pool, bytes);
        if (e != hipSuccess) {
            rc = fail(FS_ERR_HIP, "fs_create: hipMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e));
            break;
        }
        // Solver.cu:917-929: every field starts at zero (this also zeroes all padding)
        if (hipMemsetAsync(s->pool, 0, bytes, s->st) != hipSuccess) {
            rc = fail(FS_ERR_HIP, "fs_create: hipMemset failed");
            break;
        }
        if (p->pressure_solver == FS_PRESSURE_MGPCG && (rc = mg_alloc(s)) != FS_OK) break;
        if (p->flags & FS_FLAG_VORT_CONFINE) {
            const size_t nv = fsk::vort_elems(g);
            e = hipMalloc((void**)&s->vort[0], 3 * nv * sizeof(float));
            if (e != hipSuccess || hipMemsetAsync(s->vort[0], 0, 3 * nv * sizeof(float), s->st) != hipSuccess) {
                rc = fail(FS_ERR_HIP, "fs_create: vorticity arrays (%zu bytes): %s", 3 * nv * sizeof(float),
                          hipGetErrorString(e));
                break;
            }
            s->vort[1] = s->vort[0] + nv;
            s->vort[2] = s->vort[1] + nv;
            // the |omega| lattices the faces share (FS_VORT_LATTICE=0: every face evaluates its own)
            if (env.vort_lattice != 0 &&
                hipMalloc((void**)&s->vort_lattice, fsk::vort_lattice_elems(g) * sizeof(float)) != hipSuccess) {
                rc = fail(FS_ERR_HIP, "fs_create: vorticity lattices (%zu bytes)", fsk::vort_lattice_elems(g) * sizeof(float));
                break;
            }
        }
        float** slots[nfields] = {&s->f.ux, &s->f.uy, &s->f.uz, &s->f.nux, &s->f.nuy, &s->f.nuz,
                                  &s->f.rho, &s->f.nrho, &s->f.T, &s->f.nT, &s->f.p, &s->f.np,
                                  &s->f.div};
        for (int q = 0; q < nfields; ++q) *slots[q] = s->pool + (size_t)q * s->field_elems + s->origin;
        int np = fsk::residual_partials(g);
        if (hipMalloc((void**)&s->d_partial, sizeof(double) * np) != hipSuccess ||
            hipMalloc((void**)&s->d_res, sizeof(double)) != hipSuccess ||
            hipMalloc((void**)&s->d_flag, sizeof(int)) != hipSuccess ||
            hipHostMalloc((void**)&s->h_res, sizeof(double)) != hipSuccess ||
            hipHostMalloc((void**)&s->h_flag, sizeof(int)) != hipSuccess) {
            rc = fail(FS_ERR_HIP, "fs_create: auxiliary allocation failed");
            break;
        }
        (void)hipMemsetAsync(s->d_flag, 0, sizeof(int), s->st);
        // Solver.cu:993-995: T = temperature_env (ghost planes too: they hold the same constant)
        if (p->temperature_env != 0.f) {
            FsGrid all = g;
            const int lo = g.wall_lo ? 0 : g.pghost, hi = g.wall_hi ? 0 : g.pghost;
            all.nzl = g.nzl + lo + hi;
            fsk::fill_interior(s->st, all, s->f.T - (ptrdiff_t)lo * g.plane, p->temperature_env);
        }
        if (world > 1) {
            // the halo transfers are short and sit on the critical path of the next pass: give
            // their stream the highest priority so they are not queued behind the interior kernels
            // (RCCL transport only: host-staged ranks may share one GPU, and the transfers there wait
            // for the host in the middle)
            int prio_lo = 0, prio_hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
            if (env.xchg_streams >= 0) s->xchg_streams = env.xchg_streams != 0;
            s->dbg_xchg = env.xchg_debug;
            s->host_wait = env.xchg_host_wait >= 0 ? env.xchg_host_wait != 0 : unique_id == nullptr;
            s->xchg_timeout_ms = env.xchg_timeout_ms;
            if (hipStreamCreateWithPriority(&s->cs, hipStreamNonBlocking, unique_id ? prio_hi : prio_lo) != hipSuccess ||
                hipEventCreateWithFlags(&s->ev_ready, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&s->ev_done, hipEventDisableTiming) != hipSuccess) {
                rc = fail(FS_ERR_HIP, "fs_create_slab: stream/event creation failed");
                break;
            }
            if (unique_id) {
                Rccl* R = rccl();
                if (!R) {
                    rc = fail(FS_ERR_COMM, "fs_create_slab: librccl.so.1 could not be loaded");
                    break;
                }
                ncclUniqueId id;
                std::memcpy(&id, unique_id, sizeof(id));
                // FS_FLAG_TEST_SELF_PEER (tests on one GPU): a communicator of this rank alone; the
                // slab keeps its place in the decomposition but trades planes with itself, which
                // drives the RCCL calls and the stream ordering without a second device
                s->self_peer = (p->flags & FS_FLAG_TEST_SELF_PEER) != 0;
                ncclResult_t ne = R->CommInitRank(&s->comm, s->self_peer ? 1 : world, id, s->self_peer ? 0 : rank);
                if (ne != ncclSuccess) {
                    s->comm = nullptr;
                    rc = fail(FS_ERR_COMM, "fs_create_slab: ncclCommInitRank failed: %s", R->GetErrorString(ne));
                    break;
                }
            }
        }
        // all fields are zero here: let the blocked Jacobi kernel pick its tile variant on this grid
        // (a solver whose steps solve with MGPCG takes the rule instead of ~0.1 s of timing passes it
        // would not use; its smoothing passes are timed at their first use)
        if (p->pressure_solver == FS_PRESSURE_MGPCG) {
            const int keep = g.env.jacobi_tune;
            g.env.jacobi_tune = 0;
            fsk::jacobi_autotune(s->st, g, s->f.p, s->f.np, s->f.div);
            g.env.jacobi_tune = keep;
        } else {
            fsk::jacobi_autotune(s->st, g, s->f.p, s->f.np, s->f.div);
        }
        // zero state: every ghost plane already equals its neighbour's owned plane
        s->vel_ghost = g.pghost;
        s->scal_ghost_ok = true;
        s->p_ghost = g.pghost;
        if (hipStreamSynchronize(s->st) != hipSuccess || hipGetLastError() != hipSuccess) {
            rc = fail(FS_ERR_HIP, "fs_create: initialisation kernels failed");
            break;
        }
    } while (0);
    if (rc != FS_OK) {
        const std::string keep = error_text();
        fs_destroy(s);
        error_text() = keep;
        return rc;
    }
    *out = s;
    return FS_OK;
}

}  // namespace

extern "C" {

const char* fs_last_error(void) { return error_text().c_str(); }

const char* fs_version(void) { return "fluidsolver_amd 0.1 (gfx950)"; }

int fs_default_params(fs_params* p, int nx, int ny, int nz)
{
    if (!p) return fail(FS_ERR_INVALID, "fs_default_params: null params");
    std::memset(p, 0, sizeof(*p));
    p->nx = nx;
    p->ny = ny;
    p->nz = nz;
    p->dt = 0.2f;
    p->jacobi_iters = 30;
    p->scheme = FS_SCHEME_MACCORMACK;
    p->source_x = nx / 2.f;
    p->source_y = 10.f;
    p->source_z = nz / 2.f;
    p->source_radius = 7.0f;
    p->source_rho = 0.1f;
    p->source_temperature = 2.f;
    p->temperature_env = 0.f;
    p->device = -1;
    p->flags = 0;
    p->tolerance = 0.f;
    p->check_every = 8;
    p->max_sweeps = 10000;
    p->curl_strength = 0.1f;   // Solver.h:73
    p->pressure_solver = FS_PRESSURE_JACOBI;
    p->mg_level = 4;               // Solver.h:151
    p->init_smooth_steps = 2;      // Solver.h:153
    p->bottom_smooth_steps = 50;   // Solver.h:154
    p->iter_precision = 1e-3f;     // Solver.h:69
    return FS_OK;
}

int fs_destroy(fs_solver* s)
{
    if (!s) return FS_OK;
    (void)hipSetDevice(s->device);
    // every queue that may still touch a buffer drains before anything is freed (a read-back begun
    // with fs_field_host_begin and never ended is still writing its pinned buffer)
    if (s->st) (void)hipStreamSynchronize(s->st);
    if (s->cs) (void)hipStreamSynchronize(s->cs);
    if (s->copy_st) (void)hipStreamSynchronize(s->copy_st);
    if (s->comm && rccl()) (void)rccl()->CommDestroy(s->comm);
    for (auto& pe : s->ev_pending) {
        (void)hipEventDestroy(pe.a);
        (void)hipEventDestroy(pe.b);
    }
    for (hipEvent_t e : s->ev_free) (void)hipEventDestroy(e);
    for (auto& m : s->marks) (void)hipEventDestroy(m.ev);
    for (auto& w : s->halo_waits) {
        (void)hipEventDestroy(w.asked);
        (void)hipEventDestroy(w.got);
    }
    for (hipEvent_t e : s->halo_ev_free) (void)hipEventDestroy(e);
    for (fs_solver::StepGraph& g : s->graphs) (void)hipGraphExecDestroy(g.exec);
    if (s->ev_ready) (void)hipEventDestroy(s->ev_ready);
    if (s->ev_done) (void)hipEventDestroy(s->ev_done);
    for (float*& h : s->host)
        if (h) (void)hipHostFree(h);
    for (char*& b : s->h_stage)
        if (b) (void)hipHostFree(b);
    if (s->h_res) (void)hipHostFree(s->h_res);
    if (s->h_flag) (void)hipHostFree(s->h_flag);
    if (s->d_partial) (void)hipFree(s->d_partial);
    if (s->d_res) (void)hipFree(s->d_res);
    if (s->d_flag) (void)hipFree(s->d_flag);
    if (s->pool) (void)hipFree(s->pool);
    if (s->vort[0]) (void)hipFree(s->vort[0]);
    if (s->vort_lattice) (void)hipFree(s->vort_lattice);
    if (s->snap) (void)hipFree(s->snap);
    if (s->ev_snap) (void)hipEventDestroy(s->ev_snap);
    if (s->ev_copied) (void)hipEventDestroy(s->ev_copied);
    if (s->copy_st) (void)hipStreamDestroy(s->copy_st);
    if (s->cg_r) (void)hipFree(s->cg_r);
    if (s->cg_partial) (void)hipFree(s->cg_partial);
    if (s->h_dot) (void)hipHostFree(s->h_dot);
    if (s->cs) (void)hipStreamDestroy(s->cs);
    if (s->st) (void)hipStreamDestroy(s->st);
    delete s;
    return FS_OK;
}

int fs_create(const fs_params* p, fs_solver** out)
{
    return create_common(p, 0, 1, nullptr, nullptr, nullptr, out);
}

int fs_comm_unique_id(void* id)
{
    if (!id) return fail(FS_ERR_INVALID, "fs_comm_unique_id: null buffer");
    Rccl* R = rccl();
    if (!R) return fail(FS_ERR_COMM, "fs_comm_unique_id: librccl.so.1 could not be loaded");
    ncclUniqueId u;
    FS_NCCL(R->GetUniqueId(&u));
    static_assert(sizeof(u) == FS_UNIQUE_ID_BYTES, "unique id size");
    std::memcpy(id, &u, sizeof(u));
    return FS_OK;
}

int fs_create_slab(const fs_params* p, int rank, int world, const void* unique_id,
                   fs_exchange_fn fn, void* ctx, fs_solver** out)
{
    return create_common(p, rank, world, unique_id, fn, ctx, out);
}

int fs_slab_info(const fs_solver* s, int* rank, int* world, int* z0, int* nz_local)
{
    if (!s) return fail(FS_ERR_INVALID, "null solver");
    if (rank) *rank = s->rank;
    if (world) *world = s->world;
    if (z0) *z0 = s->g.z0;
    if (nz_local) *nz_local = s->g.nzl;
    return FS_OK;
}

int fs_comm_ranks(const fs_solver* s)
{
    if (!s) return -1;
    if (!s->comm) return 0;
    int n = 0;
    if (!rccl() || rccl()->CommCount(s->comm, &n) != ncclSuccess) return -1;
    return n;
}

int fs_jacobi_config(const fs_solver* s, int* variant, int* tile_rows, int* aligned_rows, int* planes_per_chunk,
                     int* nontemporal_stores)
{
    if (!s) return fail(FS_ERR_INVALID, "null solver");
    int v, r, a, lz;
    fsk::jacobi_config(s->g, &v, &r, &a, &lz);
    if (variant) *variant = v;
    if (tile_rows) *tile_rows = r;
    if (aligned_rows) *aligned_rows = a;
    if (planes_per_chunk) *planes_per_chunk = lz;
    if (nontemporal_stores) *nontemporal_stores = fsk::jacobi_nontemporal(s->g);
    return FS_OK;
}

int fs_jacobi_deep_config(const fs_solver* s, int* sweeps_per_pass, int* planes_per_chunk, int* nontemporal_stores)
{
    if (!s) return fail(FS_ERR_INVALID, "null solver");
    int t, lz, nt;
    fsk::jacobi_deep_config(s->g, &t, &lz, &nt);
    if (sweeps_per_pass) *sweeps_per_pass = t;
    if (planes_per_chunk) *planes_per_chunk = lz;
    if (nontemporal_stores) *nontemporal_stores = nt;
    return FS_OK;
}

int fs_sync(fs_solver* s)
{
    if (!s) return fail(FS_ERR_INVALID, "null solver");
    FS_HIP(hipSetDevice(s->device));
    if (s->world > 1) {
        int rc = bounded_sync(s, s->st, "fs_sync");
        if (rc) return rc;
    } else {
        FS_HIP(hipStreamSynchronize(s->st));
    }
    return check_flag(s);
}

int fs_frame(const fs_solver* s) { return s ? s->frame : -1; }

int fs_last_sweeps(const fs_solver* s) { return s ? s->last_sweeps : -1; }

int fs_cg_residual(const fs_solver* s, float* rtr)
{
    if (!s || !rtr) return fail(FS_ERR_INVALID, "fs_cg_residual: null argument");
    if (s->prm.pressure_solver != FS_PRESSURE_MGPCG) return fail(FS_ERR_STATE, "fs_cg_residual: not an MGPCG solver");
    *rtr = s->cg_rtr;
    return FS_OK;
}

size_t fs_field_size(const fs_solver* s, int field)
{
    if (!s || field < 0 || field >= FS_NUM_FIELDS) return 0;
    int dx, dy, dz;
    field_dims(s, field, dx, dy, dz);
    return (size_t)dx * dy * dz;
}

// dev = nullptr: the field's current buffer; st = nullptr: the compute stream, synchronised
static int copy_field(fs_solver* s, int field, float* host, bool to_host, float* dev = nullptr,
                      hipStream_t st = nullptr)
{
    int dx, dy, dz;
    field_dims(s, field, dx, dy, dz);
    if (!dev) dev = *field_slot(s, field);
    hipMemcpy3DParms prm;
    std::memset(&prm, 0, sizeof(prm));
    hipPitchedPtr d = make_hipPitchedPtr(dev, (size_t)s->g.pitch * sizeof(float), (size_t)dx,
                                         (size_t)(s->g.plane / s->g.pitch));
    hipPitchedPtr h = make_hipPitchedPtr(host, (size_t)dx * sizeof(float), (size_t)dx, (size_t)dy);
    prm.srcPtr = to_host ? d : h;
    prm.dstPtr = to_host ? h : d;
    prm.extent = make_hipExtent((size_t)dx * sizeof(float), (size_t)dy, (size_t)dz);
    prm.kind = to_host ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice;
    if (st) {
        FS_HIP(hipMemcpy3DAsync(&prm, st));
        return FS_OK;
    }
    FS_HIP(hipMemcpy3DAsync(&prm, s->st));
    FS_HIP(hipStreamSynchronize(s->st));
    return FS_OK;
}

static int ensure_host_buffer(fs_solver* s, int field)
{
    if (s->host[field]) return FS_OK;
    size_t bytes = fs_field_size(s, field) * sizeof(float);
    if (hipHostMalloc((void**)&s->host[field], bytes) != hipSuccess) {
        s->host[field] = nullptr;
        return fail(FS_ERR_HIP, "pinned allocation of %zu bytes failed", bytes);
    }
    return FS_OK;
}

const float* fs_field_host(fs_solver* s, int field)
{
    if (!s || field < 0 || field >= FS_NUM_FIELDS) {
        fail(FS_ERR_INVALID, "fs_field_host: bad argument");
        return nullptr;
    }
    if (hipSetDevice(s->device) != hipSuccess) {
        fail(FS_ERR_HIP, "hipSetDevice failed");
        return nullptr;
    }
    if (s->snap_field == field) {   // a copy into this buffer is in flight: finish it first
        (void)hipEventSynchronize(s->ev_copied);
        s->snap_field = -1;
    }
    if (ensure_host_buffer(s, field) != FS_OK) return nullptr;
    if (copy_field(s, field, s->host[field], true) != FS_OK) return nullptr;
    if (check_flag(s) != FS_OK) return nullptr;
    return s->host[field];
}

int fs_field_host_begin(fs_solver* s, int field)
{
    if (!s || field < 0 || field >= FS_NUM_FIELDS) return fail(FS_ERR_INVALID, "fs_field_host_begin: bad argument");
    FS_HIP(hipSetDevice(s->device));
    if (s->snap_field >= 0) return fail(FS_ERR_STATE, "fs_field_host_begin: a read-back is already in flight");
    int rc = ensure_host_buffer(s, field);
    if (rc) return rc;
    if (!s->snap) {
        FS_HIP(hipMalloc((void**)&s->snap, s->field_elems * sizeof(float)));
        FS_HIP(hipStreamCreateWithFlags(&s->copy_st, hipStreamNonBlocking));
        FS_HIP(hipEventCreateWithFlags(&s->ev_snap, hipEventDisableTiming));
        FS_HIP(hipEventCreateWithFlags(&s->ev_copied, hipEventDisableTiming));
    }
    // device-side snapshot in stream order (0.25 ms at 512^3), so that later steps may overwrite
    // the field while the snapshot crosses PCIe on the copy stream
    float* cur = *field_slot(s, field);
    const ptrdiff_t lead = (ptrdiff_t)s->origin;
    FS_HIP(hipMemcpyAsync(s->snap, cur - lead, s->field_elems * sizeof(float), hipMemcpyDeviceToDevice, s->st));
    FS_HIP(hipEventRecord(s->ev_snap, s->st));
    FS_HIP(hipStreamWaitEvent(s->copy_st, s->ev_snap, 0));
    rc = copy_field(s, field, s->host[field], true, s->snap + lead, s->copy_st);
    if (rc) return rc;
    FS_HIP(hipEventRecord(s->ev_copied, s->copy_st));
    s->snap_field = field;
    return FS_OK;
}

const float* fs_field_host_end(fs_solver* s, int field)
{
    if (!s || field < 0 || field >= FS_NUM_FIELDS || s->snap_field != field) {
        fail(FS_ERR_STATE, "fs_field_host_end: no read-back of this field in flight");
        return nullptr;
    }
    if (hipEventSynchronize(s->ev_copied) != hipSuccess) {
        fail(FS_ERR_HIP, "fs_field_host_end: hipEventSynchronize failed");
        return nullptr;
    }
    s->snap_field = -1;
    return s->host[field];
}

const float* fs_density_host(fs_solver* s) { return fs_field_host(s, FS_RHO); }

const float* fs_velocity_host(fs_solver* s, int component)
{
    if (component < 0 || component > 2) {
        fail(FS_ERR_INVALID, "fs_velocity_host: component %d", component);
        return nullptr;
    }
    return fs_field_host(s, FS_UX + component);
}

int fs_field_upload(fs_solver* s, int field, const float* src)
{
    if (!s || !src || field < 0 || field >= FS_NUM_FIELDS)
        return fail(FS_ERR_INVALID, "fs_field_upload: bad argument");
    FS_HIP(hipSetDevice(s->device));
    if (field <= FS_UZ) s->vel_ghost = 0;
    if (field <= FS_UZ || field == FS_DIV) s->div_fresh = false;
    if (field == FS_RHO || field == FS_T) s->scal_ghost_ok = false;
    if (field == FS_P) s->p_ghost = 0;
    return copy_field(s, field, const_cast<float*>(src), false);
}

int fs_field_device(fs_solver* s, int field, float** ptr, fs_layout* layout)
{
    if (!s || !ptr || field < 0 || field >= FS_NUM_FIELDS)
        return fail(FS_ERR_INVALID, "fs_field_device: bad argument");
    *ptr = *field_slot(s, field);
    // the caller may write through the pointer: forget what was derived from these fields
    if (field <= FS_UZ || field == FS_DIV) s->div_fresh = false;
    if (field <= FS_UZ) s->vel_ghost = 0;
    if (field == FS_RHO || field == FS_T) s->scal_ghost_ok = false;
    if (field == FS_P) s->p_ghost = 0;
    if (layout) {
        layout->dx = s->g.nx + (field == FS_UX);
        layout->dy = s->g.ny + (field == FS_UY);
        layout->dz = s->g.nz + (field == FS_UZ);
        layout->z0 = s->g.z0;
        layout->nz_local = s->g.nzl + ((field == FS_UZ) ? 1 : 0);
        layout->ghost = s->g.pghost;
        layout->pitch = (size_t)s->g.pitch;
        layout->plane = (size_t)s->g.plane;
    }
    return FS_OK;
}

int fs_profile_enable(fs_solver* s, int on)
{
    if (!s) return fail(FS_ERR_INVALID, "null solver");
    s->profiling = on != 0;
    return FS_OK;
}

static int drain_events(fs_solver* s)
{
    FS_HIP(hipSetDevice(s->device));
    FS_HIP(hipStreamSynchronize(s->st));
    for (auto& pe : s->ev_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, pe.a, pe.b) == hipSuccess) s->prof_ms[pe.stage] += ms;
        s->ev_free.push_back(pe.a);
        s->ev_free.push_back(pe.b);
    }
    s->ev_pending.clear();
    return FS_OK;
}

int fs_profile_read(fs_solver* s, int stage, double* ms_total, long long* launches)
{
    if (!s || stage < 0 || stage >= FS_NUM_STAGES)
        return fail(FS_ERR_INVALID, "fs_profile_read: bad argument");
    int rc = drain_events(s);
    if (rc) return rc;
    if (ms_total) *ms_total = s->prof_ms[stage];
    if (launches) *launches = s->prof_launches[stage];
    return FS_OK;
}

int fs_halo_wait(fs_solver* s, double* host_ms, double* device_ms, long long* exchanges)
{
    if (!s) return fail(FS_ERR_INVALID, "null solver");
    FS_HIP(hipSetDevice(s->device));
    int rc = drain_halo_waits(s);
    if (rc) return rc;
    if (host_ms) *host_ms = s->xchg_host_wait_ms;
    if (device_ms) *device_ms = s->xchg_dev_wait_ms;
    if (exchanges) *exchanges = s->xchg_count;
    return FS_OK;
}

int fs_profile_reset(fs_solver* s)
{
    if (!s) return fail(FS_ERR_INVALID, "null solver");
    int rc = drain_events(s);
    if (rc) return rc;
    if ((rc = drain_halo_waits(s))) return rc;
    s->xchg_host_wait_ms = s->xchg_dev_wait_ms = 0.0;
    s->xchg_count = 0;
    for (int q = 0; q < FS_NUM_STAGES; ++q) {
        s->prof_ms[q] = 0.0;
        s->prof_launches[q] = 0;
    }
    return FS_OK;
}

int fs_residual(fs_solver* s, int which, double* out)
{
    if (!s || !out || (which != 0 && which != 1))
        return fail(FS_ERR_INVALID, "fs_residual: bad argument");
    if (which == 1 && !s->prev_held)
        return fail(FS_ERR_STATE, "fs_residual: the iterate before the last sweep is not held "
                                  "(the last sweeps ran as one fused multi-sweep pass)");
    FS_HIP(hipSetDevice(s->device));
    return residual_sync(s, which == 1, out);
}

int fs_residual_noalloc(fs_solver* s, int which, double* out)
{
    if (!s || !out || (which != 0 && which != 1))
        return fail(FS_ERR_INVALID, "fs_residual_noalloc: bad argument");
    if (s->world > 1)
        return fail(FS_ERR_STATE, "fs_residual_noalloc: single-GPU solvers only (a z-slab needs a halo "
                                  "exchange and an all-reduce first)");
    if (which == 1 && !s->prev_held)
        return fail(FS_ERR_STATE, "fs_residual_noalloc: the iterate before the last sweep is not held "
                                  "(the last sweeps ran as one fused multi-sweep pass)");
    const long long rows = (long long)s->g.ny * s->g.nzl;
    if (rows > fsk::residual_static_rows())
        return fail(FS_ERR_INVALID, "fs_residual_noalloc: ny * nz = %lld rows exceed the %d rows of the static "
                                    "scratch; use fs_residual", rows, fsk::residual_static_rows());
    FS_HIP(hipSetDevice(s->device));
    double *partial = nullptr, *total = nullptr;
    if (!fsk::residual_static_scratch(s->device, &partial, &total))
        return fail(FS_ERR_STATE, "fs_residual_noalloc: no static scratch address for device %d", s->device);
    // every partial the final kernel reads is written by this call; nothing has to start at zero
    fsk::residual(s->st, s->g, which == 1 ? s->f.np : s->f.p, s->f.div, partial, total);
    FS_HIP(hipGetLastError());
    FS_HIP(hipMemcpyAsync(s->h_res, total, sizeof(double), hipMemcpyDeviceToHost, s->st));
    FS_HIP(hipStreamSynchronize(s->st));
    *out = *s->h_res;
    return FS_OK;
}

}  // extern "C"
