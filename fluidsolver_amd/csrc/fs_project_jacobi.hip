// fs_project_jacobi.hip -- the Jacobi pressure solve of a step (Solver::Project on its Jacobi
// branch, CuSolver/Source/Solver.cu:1178-1189: JacobiKernel :426-443 + swap :1183, `max_iter`
// times) and the residual norm the reference prints per sweep (:1185-1187).
//
// One domain: fused passes of several sweeps over memory (fs_jacobi.hip).  A z-slab: t fused sweeps
// need t valid ghost planes of p, because a slab face that is not a global wall loses one plane of
// validity per sweep.  Each fused pass runs as two launches: first the planes next to each
// neighbour, whose results then travel to the neighbours' ghost planes while the second launch
// produces the interior, which needs no ghost plane.  Slabs whose fields exceed the Infinity Cache
// run most of their sweeps in passes of six (k_jacobi_deep), the rest in pairs of four-sweep passes.
#include "fs_solver_impl.h"

#include <cstdio>
#include <utility>

using namespace fsi;

namespace fsi {

int run_jacobi(fs_solver* s, int sweeps)
{
    if (sweeps <= 0) return FS_OK;
    const FsGrid& g = s->g;
    const int naive = (s->prm.flags & FS_FLAG_JACOBI_NAIVE) ? 1 : 0;
    if (s->world == 1) {
        int in_np = 0, held = 0;
        {
            StageScope sc(s, FS_STAGE_JACOBI, fsk::jacobi_launches(g, sweeps, naive));
            fsk::jacobi_multi(s->st, g, s->f.p, s->f.np, s->f.div, sweeps, naive, &in_np, &held);
        }
        s->prev_held = held != 0;
        if (in_np) std::swap(s->f.p, s->f.np);
    } else {
        // slab: t fused sweeps need t valid ghost planes of p; a slab face that is not a global
        // wall loses one plane of validity per sweep, which is what the ghost planes pay for.
        // Each fused pass runs as two launches: first the T boundary planes next to each neighbour,
        // whose results then travel to the neighbours' ghost planes while the second launch
        // produces the interior, which needs no ghost plane.  So every pass has exactly one
        // exchange, hidden behind its interior, and the pressure leaves the solve with T valid
        // ghost planes (what the gradient and the next step's first pass need).
        const int T = naive ? 1 : fsk::jacobi_block();
        const int z1 = g.z0 + g.nzl;
        // (every rank must make the same sequence of exchanges, so the schedule is chosen by the
        // thinnest slab of the decomposition, not by this rank's own)
        const int thinnest = g.nz / s->world;
        const bool split = !naive && fsk::jacobi_can_block(g) && thinnest >= 2 * T + 1;
        // With ghost planes two passes deep, two passes share one exchange: pass A produces the
        // slab and T planes beyond it from the 2T valid ghost planes (redundantly with the
        // neighbour), pass B the slab itself, back in the buffer A read.  Boundary first again:
        // A then B on the planes the neighbours need (and what those depend on), the exchange of
        // the finished 2T planes, then A and B on the interior while they travel.  B's boundary
        // output overwrites planes of the input buffer that A's interior launch no longer reads.
        const bool pairs = split && g.pghost >= 2 * T && thinnest >= 6 * T + 1;
        int left = sweeps, launches = 0;
        StageScope sc(s, FS_STAGE_JACOBI, 0);
        // Six-sweep passes first, where the decomposition runs them: TD valid ghost planes in, the TD
        // (or more) boundary planes next to each neighbour first, their exchange, the interior.  The
        // last of them sends as many planes as the pass after it wants valid, so that no pass of the
        // solve has to wait for an exchange of its own.
        const int TD = fsk::jacobi_deep_block();
        const bool deep = split && fsk::jacobi_deep_slab(g);
        for (int q = deep ? fsk::jacobi_deep_count(sweeps, pairs ? 2 * T : T) : 0; q > 0; --q) {
            int rc = ensure_p_ghosts(s, TD);
            if (rc) return rc;
            const int after = left - TD;
            const int bw = (q == 1 && pairs && after >= 2 * T) ? 2 * T : TD;
            const int zin_lo = g.wall_lo ? 0 : g.z0 - TD, zin_hi = g.wall_hi ? g.nz : z1 + TD;
            const int ilo = g.wall_lo ? g.z0 : g.z0 + bw, ihi = g.wall_hi ? z1 : z1 - bw;
            FS_TRACE(s, "launch six-sweep Jacobi pass, boundary planes");
            fsk::jacobi_deep_pass(s->st, g, s->f.p, s->f.np, s->f.div, zin_lo, zin_hi, g.z0, ilo, ihi, z1);
            xmark(s, s->st, "six-sweep Jacobi pass, boundary planes");
            Seg seg = make_seg(s, s->f.np, bw, false);
            rc = exchange_begin(s, &seg, 1);
            if (rc) return rc;
            FS_TRACE(s, "launch six-sweep Jacobi pass, interior planes");
            fsk::jacobi_deep_pass(s->st, g, s->f.p, s->f.np, s->f.div, zin_lo, zin_hi, ilo, ihi);
            xmark(s, s->st, "six-sweep Jacobi pass, interior planes");
            std::swap(s->f.p, s->f.np);
            s->p_ghost = bw;
            s->prev_held = false;
            launches += 2;
            left = after;
        }
        while (pairs && left >= 2 * T) {
            int rc = ensure_p_ghosts(s, 2 * T);
            if (rc) return rc;
            float *p = s->f.p, *np = s->f.np;
            const float* div = s->f.div;
            const int inA_lo = g.wall_lo ? 0 : g.z0 - 2 * T, inA_hi = g.wall_hi ? g.nz : z1 + 2 * T;
            const int inB_lo = g.wall_lo ? 0 : g.z0 - T, inB_hi = g.wall_hi ? g.nz : z1 + T;
            // boundary ranges of A and B (empty at a global wall), then their interiors
            const int a_lo0 = g.z0 - T, a_lo1 = g.wall_lo ? a_lo0 : g.z0 + 3 * T;
            const int a_hi1 = z1 + T, a_hi0 = g.wall_hi ? a_hi1 : z1 - 3 * T;
            const int b_lo0 = g.z0, b_lo1 = g.wall_lo ? b_lo0 : g.z0 + 2 * T;
            const int b_hi1 = z1, b_hi0 = g.wall_hi ? b_hi1 : z1 - 2 * T;
            FS_TRACE(s, "launch Jacobi pair, boundary planes");
            fsk::jacobi_pass(s->st, g, p, np, div, T, inA_lo, inA_hi, a_lo0, a_lo1, a_hi0, a_hi1);
            fsk::jacobi_pass(s->st, g, np, p, div, T, inB_lo, inB_hi, b_lo0, b_lo1, b_hi0, b_hi1);
            xmark(s, s->st, "Jacobi pair, boundary planes");
            Seg seg = make_seg(s, p, 2 * T, false);
            rc = exchange_begin(s, &seg, 1);
            if (rc) return rc;
            FS_TRACE(s, "launch Jacobi pair, interior planes");
            fsk::jacobi_pass(s->st, g, p, np, div, T, inA_lo, inA_hi, g.wall_lo ? g.z0 : a_lo1, g.wall_hi ? z1 : a_hi0);
            fsk::jacobi_pass(s->st, g, np, p, div, T, inB_lo, inB_hi, g.wall_lo ? g.z0 : b_lo1, g.wall_hi ? z1 : b_hi0);
            xmark(s, s->st, "Jacobi pair, interior planes");
            s->p_ghost = 2 * T;
            s->prev_held = false;
            launches += 4;
            left -= 2 * T;
        }
        while (left > 0) {
            int t = left < T ? left : T;
            int rc = ensure_p_ghosts(s, t);
            if (rc) return rc;
            const int zin_lo = g.wall_lo ? 0 : g.z0 - t, zin_hi = g.wall_hi ? g.nz : z1 + t;
            if (split) {
                // boundary layers [z0, z0+T) and [z1-T, z1) (none at a global wall), then interior
                const int ilo = g.wall_lo ? g.z0 : g.z0 + T, ihi = g.wall_hi ? z1 : z1 - T;
                // (first on the compute stream: run beside the interior launch, which fills every CU,
                // they took 180 us instead of 25 and delayed the transfer that waits for them)
                FS_TRACE(s, "launch Jacobi boundary planes");
                fsk::jacobi_pass(s->st, g, s->f.p, s->f.np, s->f.div, t, zin_lo, zin_hi, g.z0, ilo, ihi, z1);
                xmark(s, s->st, "Jacobi pass, boundary planes");
                Seg seg = make_seg(s, s->f.np, T, false);
                rc = exchange_begin(s, &seg, 1);
                if (rc) return rc;
                FS_TRACE(s, "launch Jacobi interior planes");
                fsk::jacobi_pass(s->st, g, s->f.p, s->f.np, s->f.div, t, zin_lo, zin_hi, ilo, ihi);
                xmark(s, s->st, "Jacobi pass, interior planes");
                std::swap(s->f.p, s->f.np);
                s->p_ghost = T;
                launches += 2;
            } else {
                bool fused = !naive && fsk::jacobi_pass(s->st, g, s->f.p, s->f.np, s->f.div, t, zin_lo,
                                                        zin_hi, g.z0, z1);
                if (!fused) {
                    t = 1;
                    fsk::jacobi_sweep(s->st, g, s->f.p, s->f.np, s->f.div, g.z0, z1);
                }
                std::swap(s->f.p, s->f.np);
                s->p_ghost = 0;
                ++launches;
            }
            s->prev_held = t == 1;
            left -= t;
        }
        {
            int rc = exchange_wait(s);
            if (rc) return rc;
        }
        if (s->profiling) s->prof_launches[FS_STAGE_JACOBI] += launches;
    }
    s->last_sweeps += sweeps;
    FS_HIP(hipGetLastError());
    return FS_OK;
}

// sum r^2 over the whole grid (all ranks), binary64
int residual_sync(fs_solver* s, bool previous, double* out)
{
    if (s->world > 1) {
        if (previous)
            return fail(FS_ERR_STATE, "fs_residual(which=1) is not available on a z-slab");
        int rc = ensure_p_ghosts(s, 1);
        if (rc) return rc;
    }
    fsk::residual(s->st, s->g, previous ? s->f.np : s->f.p, s->f.div, s->d_partial, s->d_res);
    FS_HIP(hipGetLastError());
    if (s->world > 1 && s->comm) {
        Rccl* R = rccl();
        FS_NCCL(R->AllReduce(s->d_res, s->d_res, 1, ncclDouble, ncclSum, s->comm, s->st));
    }
    FS_HIP(hipMemcpyAsync(s->h_res, s->d_res, sizeof(double), hipMemcpyDeviceToHost, s->st));
    FS_HIP(hipStreamSynchronize(s->st));
    if (s->world > 1 && !s->comm) {
        int rc = s->xfn(s->xctx, FS_XCHG_SUM_F64, s->h_res, nullptr, nullptr, nullptr, sizeof(double));
        if (rc) return fail(FS_ERR_COMM, "rank %d: the all-reduce callback returned %d", s->rank, rc);
    }
    *out = *s->h_res;
    return FS_OK;
}

}  // namespace fsi
