// fs_step.hip -- one simulation step (Solver::Update -> UpdateCuda, CuSolver/Source/Solver.cu:
// 1132-1137, 1009-1114): the stages launched eagerly, or, on one GPU with a fixed sweep count, the
// replay of a hipGraph captured from those very launches.
#include "fs_solver_impl.h"

#include <cstring>

using namespace fsi;

namespace fsi {

int step_eager(fs_solver* s)
{
    int rc;
    FS_TRACE(s, "step: source, force");
    if ((rc = fs_stage_source(s))) return rc;
    if ((rc = fs_stage_force(s))) return rc;
    if ((rc = fs_stage_vorticity(s))) return rc;
    FS_TRACE(s, "step: advect");
    if ((rc = stage_advect(s, true))) return rc;
    if ((rc = fs_stage_confine(s))) return rc;
    FS_TRACE(s, "step: project");
    if ((rc = fs_stage_project(s))) return rc;
    FS_TRACE(s, "step: gradient");
    rc = fs_stage_gradient(s);
    FS_TRACE(s, "step: done");
    return rc;
}

}  // namespace fsi

namespace {

fs_solver::HostState host_state(const fs_solver* s)
{
    return {s->f, s->prev_held, s->scal_ghost_ok, s->div_fresh, s->vel_ghost, s->p_ghost};
}

void set_host_state(fs_solver* s, const fs_solver::HostState& h)
{
    s->f = h.f;
    s->prev_held = h.prev_held;
    s->vel_ghost = h.vel_ghost;
    s->scal_ghost_ok = h.scal_ghost_ok;
    s->div_fresh = h.div_fresh;
    s->p_ghost = h.p_ghost;
}

// One step as a graph replay (one GPU, fixed sweep count, no stage timing).  Which launches a step
// makes is decided by the assignment of the buffers to their roles alone.
int step_graph(fs_solver* s)
{
    FS_HIP(hipSetDevice(s->device));
    const fs_solver::HostState now = host_state(s);
    for (const fs_solver::StepGraph& g : s->graphs) {
        if (std::memcmp(&g.before.f, &now.f, sizeof(FsFields)) != 0) continue;
        FS_HIP(hipGraphLaunch(g.exec, s->st));
        // host-side state as the eager stages leave it
        set_host_state(s, g.after);
        s->last_sweeps = s->prm.jacobi_iters;
        return FS_OK;
    }
    if (s->graphs.size() >= 8) {   // an unusual sequence of stage calls and uploads: start over
        for (fs_solver::StepGraph& g : s->graphs) (void)hipGraphExecDestroy(g.exec);
        s->graphs.clear();
    }
    fs_solver::StepGraph g{};
    g.before = now;
    FS_HIP(hipStreamBeginCapture(s->st, hipStreamCaptureModeThreadLocal));
    int rc = step_eager(s);
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamEndCapture(s->st, &graph);
    if (rc == FS_OK && e == hipSuccess) {
        e = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
        if (e != hipSuccess) g.exec = nullptr;
    }
    if (graph) (void)hipGraphDestroy(graph);
    if (rc != FS_OK || e != hipSuccess) {
        // nothing ran while capturing: put the host-side state back
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        (void)hipGetLastError();
        set_host_state(s, now);
        if (rc) return rc;
        return fail(FS_ERR_HIP, "capture of a step failed: %s", hipGetErrorString(e));
    }
    g.after = host_state(s);
    s->graphs.push_back(g);
    FS_HIP(hipGraphLaunch(g.exec, s->st));
    return FS_OK;
}

}  // namespace

extern "C" {

int fs_step(fs_solver* s)
{
    if (!s) return fail(FS_ERR_INVALID, "null solver");
    // A z-slab launches eagerly: its exchanges involve the host (the host-staged transport) or a
    // communication library whose calls are not captured here.  The CG loop of an MGPCG step and
    // the convergence mode read scalars on the host, so they cannot be captured either.
    const bool graph = s->world == 1 && s->prm.tolerance <= 0.f && !s->profiling &&
                       s->prm.pressure_solver == FS_PRESSURE_JACOBI && !(s->prm.flags & FS_FLAG_NO_GRAPH);
    int rc = graph ? step_graph(s) : step_eager(s);
    if (rc) return rc;
    s->frame++;
    return FS_OK;
}

int fs_step_graphs(const fs_solver* s) { return s ? (int)s->graphs.size() : -2; }

int fs_steps(fs_solver* s, int n)
{
    for (int q = 0; q < n; ++q) {
        int rc = fs_step(s);
        if (rc) return rc;
    }
    return FS_OK;
}

}  // extern "C"
