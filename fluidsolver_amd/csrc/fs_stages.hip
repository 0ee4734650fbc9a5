// fs_stages.hip -- the stage entry points of the C ABI, in the order Solver::UpdateCuda runs them
// (CuSolver/Source/Solver.cu:1009-1114; Advect :1145-1165, Project :1168-1189), with the ghost-plane
// exchanges a z-slab needs before each of them.
#include "fs_solver_impl.h"

#include <utility>

using namespace fsi;

extern "C" {

int fs_stage_source(fs_solver* s)
{
    if (!s) return fail(FS_ERR_INVALID, "null solver");
    FS_HIP(hipSetDevice(s->device));
    FsSource src{s->prm.source_x, s->prm.source_y, s->prm.source_z, s->prm.source_radius,
                 s->prm.source_rho, s->prm.source_temperature};
    StageScope sc(s, FS_STAGE_SOURCE, 1);
    fsk::source(s->st, s->g, src, s->f.rho, s->f.T);
    s->scal_ghost_ok = false;
    FS_HIP(hipGetLastError());
    return FS_OK;
}

int fs_stage_force(fs_solver* s)
{
    if (!s) return fail(FS_ERR_INVALID, "null solver");
    FS_HIP(hipSetDevice(s->device));
    StageScope sc(s, FS_STAGE_FORCE, 1);
    fsk::force_uy(s->st, s->g, s->f.uy, s->f.T, s->prm.dt, s->prm.temperature_env);
    s->vel_ghost = 0;
    s->div_fresh = false;
    FS_HIP(hipGetLastError());
    return FS_OK;
}

int fs_stage_vorticity(fs_solver* s)
{
    if (!s) return fail(FS_ERR_INVALID, "null solver");
    if (!(s->prm.flags & FS_FLAG_VORT_CONFINE)) return FS_OK;
    FS_HIP(hipSetDevice(s->device));
    if (s->world > 1) {
        // the curl at a rank's lowest cells reads the velocity one plane below them: the exchange
        // the advection is about to ask for anyway
        int rc = ensure_advect_ghosts(s);
        if (rc) return rc;
    }
    StageScope sc(s, FS_STAGE_VORTICITY, 1);
    fsk::find_vort(s->st, s->g, s->vort[0], s->vort[1], s->vort[2], s->f.ux, s->f.uy, s->f.uz);
    FS_HIP(hipGetLastError());
    if (s->world > 1) {
        // fill the halo planes of the windows: the first / last h planes of every rank's own entries
        // go to the lower / upper neighbour (planes of nx*ny entries; see fs_vort.hip)
        const FsGrid& g = s->g;
        const size_t cell = (size_t)g.nx * g.ny;
        const int h = fsk::vort_halo_planes(g);
        const size_t own = (size_t)(g.wall_lo ? 0 : h) * cell;   // offset of this rank's first own entry
        Seg segs[3];
        for (int q = 0; q < 3; ++q) {
            float* base = s->vort[q] + own;
            segs[q].n = (size_t)h * cell;
            segs[q].send_lo = g.wall_lo ? nullptr : base;
            segs[q].recv_lo = g.wall_lo ? nullptr : s->vort[q];
            segs[q].send_hi = g.wall_hi ? nullptr : base + (size_t)(g.nzl - h) * cell;
            segs[q].recv_hi = g.wall_hi ? nullptr : base + (size_t)g.nzl * cell;
        }
        int rc = exchange(s, segs, 3);
        if (rc) return rc;
    }
    return FS_OK;
}

int fs_stage_confine(fs_solver* s)
{
    if (!s) return fail(FS_ERR_INVALID, "null solver");
    if (!(s->prm.flags & FS_FLAG_VORT_CONFINE)) return FS_OK;
    FS_HIP(hipSetDevice(s->device));
    StageScope sc(s, FS_STAGE_CONFINE, 3);
    fsk::vort_confine(s->st, s->g, s->f.ux, s->f.uy, s->f.uz, s->vort[0], s->vort[1], s->vort[2],
                      s->prm.curl_strength, s->vort_lattice);
    s->vel_ghost = 0;
    s->div_fresh = false;
    FS_HIP(hipGetLastError());
    return FS_OK;
}

}  // extern "C"

namespace fsi {

// fuse_div: let the scalar advection also write the divergence of the new velocity (only inside
// fs_step, where the projection follows at once; a stand-alone fs_stage_advect leaves div alone,
// as Solver::Advect does)
int stage_advect(fs_solver* s, bool fuse_div)
{
    FS_HIP(hipSetDevice(s->device));
    FsFields& f = s->f;
    const float dt = s->prm.dt;
    const int sc = s->prm.scheme;
    const FsGrid& g = s->g;
    int rc;
    // A slab hides each ghost exchange behind the planes that cannot need it: a back-trace stays
    // within `ghost` planes (or raises the halo flag), so the planes at least that far from a
    // neighbour are produced while the ghost planes travel, the others after they have arrived.
    const int m_lo = g.wall_lo ? 0 : g.ghost, m_hi = g.wall_hi ? 0 : g.ghost;
    const bool split = s->world > 1 && g.nzl >= m_lo + m_hi + 4;
    const int z0 = g.z0, z1 = g.z0 + g.nzl;
    // Solver.cu:1153-1158: all three components from the OLD velocity, then swap
    if (!split) {
        if ((rc = ensure_advect_ghosts(s))) return rc;
        StageScope ps(s, FS_STAGE_ADVECT_VELOCITY, fsk::advect_velocity_launches(g));
        fsk::advect_velocity(s->st, g, f.nux, f.nuy, f.nuz, f.ux, f.uy, f.uz, dt, sc, s->d_flag);
    } else {
        StageScope ps(s, FS_STAGE_ADVECT_VELOCITY, 3 * fsk::advect_velocity_launches(g));
        if ((rc = advect_ghosts_begin(s))) return rc;
        fsk::advect_velocity(s->st, g, f.nux, f.nuy, f.nuz, f.ux, f.uy, f.uz, dt, sc, s->d_flag, z0 + m_lo,
                             z1 - m_hi + (g.wall_hi ? 1 : 0), true);
        if ((rc = exchange_wait(s))) return rc;
        if (s->vel_ghost < g.ghost) s->vel_ghost = g.ghost;
        s->scal_ghost_ok = true;
        fsk::advect_velocity(s->st, g, f.nux, f.nuy, f.nuz, f.ux, f.uy, f.uz, dt, sc, s->d_flag, z0, z0 + m_lo);
        if (!g.wall_hi)
            fsk::advect_velocity(s->st, g, f.nux, f.nuy, f.nuz, f.ux, f.uy, f.uz, dt, sc, s->d_flag, z1 - m_hi,
                                 z1 + 1);
    }
    std::swap(f.ux, f.nux);
    std::swap(f.uy, f.nuy);
    std::swap(f.uz, f.nuz);
    s->vel_ghost = 0;
    // Solver.cu:1160-1164: T then rho with the NEW (advected, unprojected) velocity.  The kernel
    // also leaves the divergence of the new velocity behind for the owned cells, unless the
    // confinement force is still to change that velocity.
    float* div = (fuse_div && !(s->prm.flags & FS_FLAG_VORT_CONFINE)) ? f.div : nullptr;
    if (!split) {
        // (the same planes a splitting neighbour trades at this point: pghost of them)
        if ((rc = ensure_velocity_ghosts(s, g.pghost))) return rc;
        StageScope ps(s, FS_STAGE_ADVECT_SCALARS, 1);
        s->div_fresh = fsk::advect_scalars(s->st, g, f.T, f.nT, f.rho, f.nrho, f.ux, f.uy, f.uz, dt, sc,
                                           s->d_flag, div);
    } else {
        StageScope ps(s, FS_STAGE_ADVECT_SCALARS, 3);
        // (pghost planes: the divergence the pressure passes need beyond the slab comes from them)
        Seg segs[3] = {make_seg(s, f.ux, g.pghost, false), make_seg(s, f.uy, g.pghost, false),
                       make_seg(s, f.uz, g.pghost, true)};
        if ((rc = exchange_begin(s, segs, 3))) return rc;
        bool fresh = fsk::advect_scalars(s->st, g, f.T, f.nT, f.rho, f.nrho, f.ux, f.uy, f.uz, dt, sc, s->d_flag,
                                         div, z0 + m_lo, z1 - m_hi, true);
        if ((rc = exchange_wait(s))) return rc;
        s->vel_ghost = g.pghost;
        fresh &= fsk::advect_scalars(s->st, g, f.T, f.nT, f.rho, f.nrho, f.ux, f.uy, f.uz, dt, sc, s->d_flag, div,
                                     z0, z0 + m_lo);
        fresh &= fsk::advect_scalars(s->st, g, f.T, f.nT, f.rho, f.nrho, f.ux, f.uy, f.uz, dt, sc, s->d_flag, div,
                                     z1 - m_hi, z1);
        s->div_fresh = fresh;
    }
    std::swap(f.T, f.nT);
    std::swap(f.rho, f.nrho);
    s->scal_ghost_ok = false;
    FS_HIP(hipGetLastError());
    return FS_OK;
}

}  // namespace fsi

extern "C" {

int fs_stage_advect(fs_solver* s)
{
    if (!s) return fail(FS_ERR_INVALID, "null solver");
    return stage_advect(s, false);
}

int fs_stage_divergence(fs_solver* s)
{
    if (!s) return fail(FS_ERR_INVALID, "null solver");
    FS_HIP(hipSetDevice(s->device));
    const FsGrid& g = s->g;
    int zb = g.z0, ze = g.z0 + g.nzl;
    if (s->world > 1) {
        // the fused Jacobi passes recompute sweeps on the ghost planes of the pressure, so div is
        // needed pghost - 1 planes beyond the slab; the velocity ghosts provide it
        int rc = ensure_velocity_ghosts(s, g.pghost);
        if (rc) return rc;
        const int ext = g.pghost - 1;
        if (!g.wall_lo) zb -= ext;
        if (!g.wall_hi) ze += ext;
    }
    if (s->div_fresh) {
        // the owned cells were written by the scalar advection from the same velocity; a slab
        // still needs the planes beyond it
        if (s->world == 1) return FS_OK;
        StageScope sc(s, FS_STAGE_DIVERGENCE, 2);
        fsk::divergence(s->st, g, s->f.div, s->f.ux, s->f.uy, s->f.uz, zb, g.z0);
        fsk::divergence(s->st, g, s->f.div, s->f.ux, s->f.uy, s->f.uz, g.z0 + g.nzl, ze);
        FS_HIP(hipGetLastError());
        return FS_OK;
    }
    StageScope sc(s, FS_STAGE_DIVERGENCE, 1);
    fsk::divergence(s->st, g, s->f.div, s->f.ux, s->f.uy, s->f.uz, zb, ze);
    s->div_fresh = true;
    FS_HIP(hipGetLastError());
    return FS_OK;
}

int fs_stage_jacobi(fs_solver* s, int sweeps)
{
    if (!s) return fail(FS_ERR_INVALID, "null solver");
    if (sweeps < 0) return fail(FS_ERR_INVALID, "fs_stage_jacobi: sweeps < 0");
    FS_HIP(hipSetDevice(s->device));
    s->last_sweeps = 0;
    return run_jacobi(s, sweeps);
}

int fs_stage_project(fs_solver* s)
{
    if (!s) return fail(FS_ERR_INVALID, "null solver");
    int rc = fs_stage_divergence(s);
    if (rc) return rc;
    return fs_stage_pressure_solve(s);
}

int fs_stage_pressure_solve(fs_solver* s)
{
    if (!s) return fail(FS_ERR_INVALID, "null solver");
    FS_HIP(hipSetDevice(s->device));
    int rc;
    s->last_sweeps = 0;
    if (s->prm.pressure_solver == FS_PRESSURE_MGPCG) return s->world > 1 ? run_mgpcg_slab(s) : run_mgpcg(s);
    if (s->prm.tolerance > 0.f) {
        // convergence mode: groups of check_every sweeps until sum r^2 < tolerance
        while (s->last_sweeps < s->prm.max_sweeps) {
            int n = s->prm.check_every;
            if (s->last_sweeps + n > s->prm.max_sweeps) n = s->prm.max_sweeps - s->last_sweeps;
            rc = run_jacobi(s, n);
            if (rc) return rc;
            double r2 = 0.0;
            rc = residual_sync(s, false, &r2);
            if (rc) return rc;
            if (r2 < (double)s->prm.tolerance) break;
        }
        return FS_OK;
    }
    return run_jacobi(s, s->prm.jacobi_iters);
}

int fs_stage_gradient(fs_solver* s)
{
    if (!s) return fail(FS_ERR_INVALID, "null solver");
    FS_HIP(hipSetDevice(s->device));
    int rc = ensure_p_ghosts(s, 1);
    if (rc) return rc;
    StageScope sc(s, FS_STAGE_GRADIENT, 1);
    fsk::gradient(s->st, s->g, s->f.ux, s->f.uy, s->f.uz, s->f.p);
    s->vel_ghost = 0;
    s->div_fresh = false;
    FS_HIP(hipGetLastError());
    return FS_OK;
}

int fs_mg_vcycle(fs_solver* s)
{
    if (!s) return fail(FS_ERR_INVALID, "null solver");
    if (s->prm.pressure_solver != FS_PRESSURE_MGPCG)
        return fail(FS_ERR_STATE, "fs_mg_vcycle: not an MGPCG solver");
    FS_HIP(hipSetDevice(s->device));
    return mg_vcycle(s);
}

int fs_field_fill(fs_solver* s, int field, float value)
{
    if (!s || field < 0 || field >= FS_NUM_FIELDS) return fail(FS_ERR_INVALID, "fs_field_fill: bad argument");
    FS_HIP(hipSetDevice(s->device));
    float* base = nullptr;
    switch (field) {
    case FS_UX: base = s->f.ux; break;
    case FS_UY: base = s->f.uy; break;
    case FS_UZ: base = s->f.uz; break;
    case FS_RHO: base = s->f.rho; break;
    case FS_T: base = s->f.T; break;
    case FS_P: base = s->f.p; break;
    default: base = s->f.div; break;
    }
    // the field's own extent only: what lies beyond it in a row, a plane or the array stays zero
    FsGrid g = s->g;
    g.nx += field == FS_UX;
    g.ny += field == FS_UY;
    g.nzl += field == FS_UZ;
    fsk::fill_interior(s->st, g, base, value);
    FS_HIP(hipGetLastError());
    if (field <= FS_UZ) s->vel_ghost = 0;
    if (field <= FS_UZ || field == FS_DIV) s->div_fresh = false;
    if (field == FS_RHO || field == FS_T) s->scal_ghost_ok = false;
    if (field == FS_P) {
        s->p_ghost = 0;
        s->prev_held = false;
    }
    return FS_OK;
}

}  // extern "C"
