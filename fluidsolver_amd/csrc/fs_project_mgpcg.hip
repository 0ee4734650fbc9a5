// fs_project_mgpcg.hip -- the multigrid-preconditioned conjugate gradient solve of a step, the
// branch of Solver::Project the reference ships (Conjugate, CuSolver/Source/Solver.cu:1226-1297;
// MG_Preconditioner :1312-1366): host-side control flow on one domain and on a z-slab's windows of
// the levels.  Kernels and per-level launchers: fs_mgpcg.hip.
#include "fs_solver_impl.h"

#include <cstdio>
#include <utility>

using namespace fsi;

namespace {

// the sum a fused kernel pair has left in cg_out, as the float the host logic reads (synchronises)
int cg_read_sum(fs_solver* s, float* out)
{
    FS_HIP(hipMemcpyAsync(s->h_dot, s->cg_out, sizeof(double), hipMemcpyDeviceToHost, s->st));
    FS_HIP(hipStreamSynchronize(s->st));
    *out = (float)*s->h_dot;
    return FS_OK;
}

}  // namespace

namespace fsi {

// Work arrays of the solve, one allocation, zeroed on the solver's stream: r, z and a second z over
// all levels, then p, Ap, x (and on one domain a second p) on the fine grid; the dot-product scratch.
// A z-slab holds a dense window of every level instead: its owned planes plus G ghost planes per
// non-wall side.  G planes last G colours of smoothing between two exchanges, so the coarse levels,
// whose planes are small and whose sweeps are many, get deeper ghosts than the fine ones (a level
// with large planes: 4, what one blocked pass uses up); never deeper than the slab of the level, nor
// than twice the next coarser level's (prolongation halves the depth).  FS_MG_SLAB_GHOST caps every
// level (tests).
int mg_alloc(fs_solver* s)
{
    const FsGrid& g = s->g;
    const int L = s->prm.mg_level;
    size_t space_al, n;
    int fine_arrays;
    if (s->world > 1) {
        const int cap_all = g.env.mg_slab_ghost >= 1 ? g.env.mg_slab_ghost : 1 << 20;
        int Gs[16];
        for (int l = L - 1; l >= 0; --l) {
            const long plane = (long)(g.nx >> l) * (g.ny >> l);
            int G = plane >= 128L * 128L ? 4 : plane >= 64L * 64L ? 8 : 16;
            if (G > (g.nzl >> l)) G = g.nzl >> l;
            if (l < L - 1 && G > 2 * Gs[l + 1]) G = 2 * Gs[l + 1];
            // level 0 starts as r = -div on every held plane, and the divergence field has
            // pghost ghost planes: its window must not reach beyond them
            if (l == 0 && G > 4) G = 4;
            if (l == 0 && G > g.pghost) G = g.pghost;
            if (G > cap_all) G = cap_all;
            Gs[l] = G;
        }
        size_t off = 0;
        for (int l = 0; l < L; ++l) {
            fs_solver::MgWin W{};
            W.nx = g.nx >> l; W.ny = g.ny >> l; W.nz = g.nz >> l;
            W.z0 = g.z0 >> l; W.nzl = g.nzl >> l;
            W.G = Gs[l];
            W.lo = g.wall_lo ? 0 : W.G; W.hi = g.wall_hi ? 0 : W.G;
            W.off = off;
            W.vz = 0;
            W.tb = -1;
            off += (W.cell() * W.held() + 63) & ~(size_t)63;
            s->mgw.push_back(W);
        }
        space_al = off;
        n = (s->mgw[0].cell() * s->mgw[0].held() + 63) & ~(size_t)63;
        fine_arrays = 3;
    } else {
        const size_t space = fsk::mg_space(g, L);
        space_al = (space + 63) & ~(size_t)63;   // keeps every array 256-byte aligned
        n = (size_t)g.nx * g.ny * g.nz;
        fine_arrays = 4;
    }
    const size_t total = (3 * space_al + fine_arrays * n) * sizeof(float);
    if (hipMalloc((void**)&s->cg_r, total) != hipSuccess || hipMemsetAsync(s->cg_r, 0, total, s->st) != hipSuccess ||
        hipMalloc((void**)&s->cg_partial, (fsk::cg_dot_partials() + 4) * sizeof(double)) != hipSuccess ||
        hipHostMalloc((void**)&s->h_dot, sizeof(double)) != hipSuccess)
        return fail(FS_ERR_HIP, "fs_create: MGPCG work arrays (%zu bytes)", total);
    s->cg_z = s->cg_r + space_al;
    s->cg_zalt = s->cg_z + space_al;
    s->cg_p = s->cg_zalt + space_al;
    s->cg_Ap = s->cg_p + n;
    s->cg_x = s->cg_Ap + n;
    if (fine_arrays == 4) s->cg_p2 = s->cg_x + n;
    s->cg_out = s->cg_partial + fsk::cg_dot_partials();
    return FS_OK;
}

// Conjugate(f_pressure, f_div), Solver.cu:1226-1297, prints left out
int run_mgpcg(fs_solver* s)
{
    const FsGrid& g = s->g;
    const fs_params& P = s->prm;
    StageScope sc(s, FS_STAGE_MGPCG, 1);
    float *r = s->cg_r, *z = s->cg_z, *p = s->cg_p, *Ap = s->cg_Ap, *x = s->cg_x;
    auto precondition = [&] {   // the result may land in either z buffer: follow it
        float* res = fsk::mg_precondition(s->st, g, r, z, s->cg_zalt, P.mg_level, P.init_smooth_steps,
                                          P.bottom_smooth_steps);
        if (res != z) {
            s->cg_zalt = z;
            s->cg_z = z = res;
        }
    };
    int rc;
    // The dot products stay on the device: alpha and beta are formed by the kernels that use them
    // (CgScalar), and the host reads back only r.r, the one number its control flow depends on.
    double *d_rtr = s->cg_out, *d_pap = s->cg_out + 1, *d_zo = s->cg_out + 2, *d_zn = s->cg_out + 3;
    fsk::cg_init(s->st, g, r, x, s->f.div);                       // :1233
    float rTr = 0.f;
    if (P.jacobi_iters <= 0) {
        // init_rTr :1237-1239 is only printed by the reference; here it is the reported r.r of a
        // solve that is not allowed a single iteration
        fsk::cg_dot(s->st, g, r, r, s->cg_partial, d_rtr);
        if ((rc = cg_read_sum(s, &rTr))) return rc;
    }
    precondition();                                               // :1244
    fsk::cg_dot(s->st, g, z, r, s->cg_partial, d_zo);             // old_zTr :1252-1254
    // p = z + beta p (:1250 with beta = 0, :1290 afterwards) is formed by the operator pass that
    // needs it, into the other of two p arrays
    float* p2 = s->cg_p2;
    fsk::CgScalar beta(0.f);
    if (P.jacobi_iters <= 0) fsk::cg_update_p(s->st, g, p, z, beta);
    int it = 0;
    for (; it < P.jacobi_iters; ++it) {
        // :1250 / :1290, :1259 and the products of :1261
        fsk::cg_update_p_apply_a_dot(s->st, g, Ap, p2, p, z, beta, s->cg_partial, d_pap);
        std::swap(p, p2);
        // alpha = old_zTr / pAp :1263, then :1266 and :1269-1271
        fsk::cg_update_xr_dot(s->st, g, x, r, p, Ap, fsk::CgScalar(d_zo, d_pap), s->cg_partial, d_rtr);
        if ((rc = cg_read_sum(s, &rTr))) return rc;
        if (rTr < P.iter_precision || rTr == 0) {                 // :1275
            ++it;
            break;
        }
        precondition();                                           // :1279
        fsk::cg_dot(s->st, g, z, r, s->cg_partial, d_zn);         // new_zTr :1285-1287
        beta = fsk::CgScalar(d_zn, d_zo);                         // new_zTr / old_zTr :1288, used by the next pass
        std::swap(d_zo, d_zn);
    }
    s->cg_p = p;
    s->cg_p2 = p2;
    s->last_sweeps = it;
    s->cg_rtr = rTr;
    // swap(&res, &x) at :1295 exchanges the by-value argument with the member x: the first solve
    // leaves the pressure field alone and makes x the pressure buffer; from then on the solution
    // accumulates in the pressure field itself
    if (s->cg_x_is_pressure) fsk::cg_to_pressure(s->st, g, s->f.p, x);
    s->cg_x_is_pressure = true;
    s->prev_held = false;
    s->p_ghost = 0;
    FS_HIP(hipGetLastError());
    return FS_OK;
}

}  // namespace fsi

namespace {

// ---- MGPCG on z-slabs ------------------------------------------------------------------------------
// Every kernel runs on the rank's whole window of a level, ghost planes included, and reads zero
// beyond it: exact at a global wall, wrong at a slab face, where the error enters the outermost
// ghost plane and moves one plane inward per stencil application.  The driver counts the ghost
// planes that are still exact (MgWin::vz) and refreshes them from the neighbours' owned planes
// before an owned plane could be touched.  The count follows the same rule on every rank (walls
// are not special-cased), so all ranks make the same sequence of exchanges.

// *slot (a device double a kernel pair has just written) becomes its sum over all ranks, on the
// device; with `out` the host also gets it, narrowed to the float the reference's host logic works
// with (synchronises).  The host-staged transport sums on the host in any case.
int cg_sum_all(fs_solver* s, double* slot, float* out)
{
    if (s->comm) {
        Rccl* R = rccl();
        FS_NCCL(R->AllReduce(slot, slot, 1, ncclDouble, ncclSum, s->comm, s->st));
        if (!out) return FS_OK;
    }
    FS_HIP(hipMemcpyAsync(s->h_dot, slot, sizeof(double), hipMemcpyDeviceToHost, s->st));
    FS_HIP(hipStreamSynchronize(s->st));
    if (!s->comm) {
        int rc = s->xfn(s->xctx, FS_XCHG_SUM_F64, s->h_dot, nullptr, nullptr, nullptr, sizeof(double));
        if (rc) return fail(FS_ERR_COMM, "rank %d: the all-reduce callback returned %d", s->rank, rc);
        // (the copy reads h_dot in stream order before the next read-back overwrites it)
        FS_HIP(hipMemcpyAsync(slot, s->h_dot, sizeof(double), hipMemcpyHostToDevice, s->st));
    }
    if (out) *out = (float)*s->h_dot;
    return FS_OK;
}

// refresh `planes` ghost planes of a window array (base = its first held plane) of level l
int mg_exchange(fs_solver* s, float* base, int l, int planes)
{
    const fs_solver::MgWin& W = s->mgw[l];
    const size_t c = W.cell();
    Seg g;
    g.n = (size_t)planes * c;
    float* own = base + (size_t)W.lo * c;
    g.send_lo = W.lo ? own : nullptr;
    g.recv_lo = W.lo ? own - (size_t)planes * c : nullptr;
    g.send_hi = W.hi ? own + (size_t)(W.nzl - planes) * c : nullptr;
    g.recv_hi = W.hi ? own + (size_t)W.nzl * c : nullptr;
    // nothing runs beside these transfers (the next launch needs them), so they go on the compute
    // stream itself: no event hand-overs between two queues for a message of a few planes
    const int keep = s->xchg_streams;
    s->xchg_streams = 0;
    const int rc = exchange(s, &g, 1);
    s->xchg_streams = keep;
    return rc;
}

// `steps` red/black sweeps of level l (Smooth, Solver.cu:848-860) on the window whose current
// iterate is in z.  Levels with large planes take four (or two) colours per launch through the
// temporally blocked kernel, from z into zalt (the two are swapped on return); the others one launch
// per colour in place.  Either way a launch of c colours uses up c exact ghost planes.
int mg_smooth_slab(fs_solver* s, int l, int steps, bool from_zero, float*& z, float*& zalt)
{
    fs_solver::MgWin& W = s->mgw[l];
    float* r = s->cg_r + W.off;
    // parity of the global index of the window's plane 0: it enters the colour of a cell, and it
    // differs between ranks, so nothing that shapes the sequence of exchanges may depend on it
    const int kpar = (W.z0 - W.lo) & 1, phases = 2 * steps;
    bool blocked = W.G >= 2 && W.tb != 0;
    for (int ph = 0; ph < phases;) {
        int take = 1;
        if (blocked) take = (phases - ph >= 4 && W.G >= 4) ? 4 : 2;
        const bool zero_in = from_zero && ph == 0;
        if (zero_in) {
            W.vz = W.G + 1;   // z = 0 is exact everywhere, and so is the first colour computed from it
        } else if (W.vz < take) {
            int rc = mg_exchange(s, z, l, W.G);
            if (rc) return rc;
            W.vz = W.G;
        }
        if (blocked && fsk::mgs_smooth_pass(s->st, s->g.env, r, z, zalt, W.nx, W.ny, W.held(), take, zero_in, kpar)) {
            std::swap(z, zalt);
            W.tb = 1;
        } else {
            if (blocked) W.tb = 0;   // (a matter of the level's shape: the same answer on every rank, every time)
            blocked = false;
            take = 1;
            fsk::mgs_smooth_phase(s->st, r, z, W.nx, W.ny, W.held(), ph, kpar);
        }
        W.vz -= take;
        ph += take;
    }
    return FS_OK;
}

// one V-cycle (MG_Preconditioner, Solver.cu:1312-1366): level 0 of r (ghost planes exact) -> level 0
// of z (s->cg_z on return; the blocked passes alternate between cg_z and cg_zalt)
int mg_precondition_slab(fs_solver* s)
{
    const fs_params& P = s->prm;
    const int L = P.mg_level;
    // z of every level starts at zero
    {
        const fs_solver::MgWin& W0 = s->mgw[0];
        const size_t n0 = W0.cell() * W0.held(), space = s->mgw[L - 1].off + s->mgw[L - 1].cell() * s->mgw[L - 1].held();
        FS_HIP(hipMemsetAsync(s->cg_z, 0, space * sizeof(float), s->st));
        (void)n0;   // (the coarse levels of r need no clearing: restriction and exchange write every held plane)
    }
    float *z[16], *za[16];
    for (int l = 0; l < L; ++l) {
        z[l] = s->cg_z + s->mgw[l].off;
        za[l] = s->cg_zalt + s->mgw[l].off;
    }
    int rc;
    for (int l = 0; l < L - 1; ++l) {
        fs_solver::MgWin &W = s->mgw[l], &C = s->mgw[l + 1];
        if ((rc = mg_smooth_slab(s, l, P.init_smooth_steps << l, true, z[l], za[l]))) return rc;
        // the residual of a child needs z one plane beyond it
        if (W.vz < 1) {
            if ((rc = mg_exchange(s, z[l], l, W.G))) return rc;
            W.vz = W.G;
        }
        fsk::mgs_restrict(s->st, s->cg_r + W.off, z[l], W.nx, W.ny, W.held(), W.lo,
                          s->cg_r + C.off + (size_t)C.lo * C.cell(), C.nzl);
        // the coarse right-hand side of the ghost planes comes from the ranks that own them
        if ((rc = mg_exchange(s, s->cg_r + C.off, l + 1, C.G))) return rc;
    }
    if ((rc = mg_smooth_slab(s, L - 1, P.bottom_smooth_steps, true, z[L - 1], za[L - 1]))) return rc;
    for (int l = L - 2; l >= 0; --l) {
        fs_solver::MgWin &W = s->mgw[l], &C = s->mgw[l + 1];
        fsk::mgs_prolongate(s->st, z[l], W.nx, W.ny, W.held(), W.z0 - W.lo, z[l + 1], C.z0 - C.lo, C.held());
        // a fine ghost plane d planes out has its parent ceil(d/2) planes out
        W.vz = W.vz < 2 * C.vz ? W.vz : 2 * C.vz;
        if ((rc = mg_smooth_slab(s, l, P.init_smooth_steps << l, false, z[l], za[l]))) return rc;
    }
    if (z[0] != s->cg_z) std::swap(s->cg_z, s->cg_zalt);
    FS_HIP(hipGetLastError());
    return FS_OK;
}

}  // namespace

namespace fsi {

// Conjugate(f_pressure, f_div), Solver.cu:1226-1297, on this rank's slab
int run_mgpcg_slab(fs_solver* s)
{
    const FsGrid& g = s->g;
    const fs_params& P = s->prm;
    StageScope sc(s, FS_STAGE_MGPCG, 1);
    const fs_solver::MgWin& W = s->mgw[0];
    const size_t cell = W.cell(), own = (size_t)W.lo * cell;
    float *r = s->cg_r, *z = s->cg_z, *p = s->cg_p, *Ap = s->cg_Ap, *x = s->cg_x;
    // the window as a grid of its own (what the element-wise and operator kernels see), and its
    // owned planes (what the dot products sum over)
    FsGrid gw = g, go = g;
    gw.nz = W.held();
    go.nz = W.nzl;
    // the dot products stay on the device, as in run_mgpcg; the host reads r.r only
    double *d_rtr = s->cg_out, *d_pap = s->cg_out + 1, *d_zo = s->cg_out + 2, *d_zn = s->cg_out + 3;
    auto dot = [&](const float* a, const float* b, double* slot, float* out) {
        fsk::cg_dot(s->st, go, a + own, b + own, s->cg_partial, slot);
        return cg_sum_all(s, slot, out);
    };
    int rc = FS_OK;
    // r = -div on every held plane (the divergence is exact pghost - 1 planes beyond the slab), x = 0
    fsk::cg_init(s->st, gw, r, x, s->f.div - (ptrdiff_t)W.lo * g.plane);               // :1233
    if (g.pghost - 1 < W.G && (rc = mg_exchange(s, r, 0, W.G))) return rc;             // (thin slabs: fewer exact div planes)
    float rTr = 0.f;
    if (P.jacobi_iters <= 0 && (rc = dot(r, r, d_rtr, &rTr))) return rc;               // :1237-1239
    if ((rc = mg_precondition_slab(s))) return rc;                                     // :1244
    z = s->cg_z;   // (the V-cycle leaves its result in either z buffer and makes that one cg_z)
    fsk::cg_update_p(s->st, gw, p, z, 0.f);                                            // :1250
    if ((rc = dot(z, r, d_zo, nullptr))) return rc;                                    // :1252-1254
    int it = 0;
    for (; it < P.jacobi_iters; ++it) {
        if ((rc = mg_exchange(s, p, 0, 1))) return rc;        // A p reaches one plane into the neighbours
        // :1259 and the products of :1261 over the owned planes
        fsk::cg_apply_a_dot(s->st, gw, Ap, p, s->cg_partial, d_pap, W.lo, W.lo + W.nzl);
        if ((rc = cg_sum_all(s, d_pap, nullptr))) return rc;
        // :1263, :1266 and :1269-1271
        fsk::cg_update_xr_dot(s->st, gw, x, r, p, Ap, fsk::CgScalar(d_zo, d_pap), s->cg_partial, d_rtr, W.lo,
                              W.lo + W.nzl);
        if ((rc = cg_sum_all(s, d_rtr, &rTr))) return rc;
        if (rTr < P.iter_precision || rTr == 0) {                                      // :1275
            ++it;
            break;
        }
        if ((rc = mg_exchange(s, r, 0, W.G))) return rc;      // the V-cycle starts from exact ghost planes
        if ((rc = mg_precondition_slab(s))) return rc;                                 // :1279
        z = s->cg_z;
        if ((rc = dot(z, r, d_zn, nullptr))) return rc;                                // :1285-1287
        fsk::cg_update_p(s->st, gw, p, z, fsk::CgScalar(d_zn, d_zo));                  // :1288-1290
        std::swap(d_zo, d_zn);
    }
    s->last_sweeps = it;
    s->cg_rtr = rTr;
    // the pointer quirk of :1295, as in run_mgpcg
    if (s->cg_x_is_pressure) fsk::cg_to_pressure(s->st, go, s->f.p, x + own);
    s->cg_x_is_pressure = true;
    s->prev_held = false;
    s->p_ghost = 0;
    FS_HIP(hipGetLastError());
    return FS_OK;
}

// One application of the preconditioner to the solver's residual array, as the reference's public
// MG_Preconditioner() does on its members r and z (Solver.cu:1312-1366): z = M^-1 r.
int mg_vcycle(fs_solver* s)
{
    if (s->world > 1) {
        int rc = mg_exchange(s, s->cg_r, 0, s->mgw[0].G);   // the V-cycle starts from exact ghost planes
        return rc ? rc : mg_precondition_slab(s);
    }
    float* res = fsk::mg_precondition(s->st, s->g, s->cg_r, s->cg_z, s->cg_zalt, s->prm.mg_level,
                                      s->prm.init_smooth_steps, s->prm.bottom_smooth_steps);
    if (res != s->cg_z) std::swap(s->cg_z, s->cg_zalt);
    FS_HIP(hipGetLastError());
    return FS_OK;
}

}  // namespace fsi
