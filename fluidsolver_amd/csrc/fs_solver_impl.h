// fs_solver_impl.h -- what the translation units behind the C ABI of include/fluidsolver.h share:
// the solver object, the error convention, the RCCL entry points and the internal interfaces
// between
//   fs_solver.hip          lifetime, buffers, getters, stage timing, residual entry points
//   fs_exchange.hip        halo exchange of a z-slab (RCCL send/recv or host-staged), ghost validity
//   fs_project_jacobi.hip  the Jacobi solve: single domain and the slab schedule, residual norm
//   fs_project_mgpcg.hip   the MGPCG solve: single domain and slab windows
//   fs_stages.hip          the stage entry points in the order of Solver::UpdateCuda
//   fs_step.hip            one step: eager launches or the replay of a captured hipGraph
// Not part of the installed interface.
#pragma once
#include "../../include/fluidsolver.h"
#include "fs_kernels.h"

#include <rccl/rccl.h>

#include <string>
#include <vector>

namespace fsi {

// stores the message fs_last_error() returns (thread-local) and hands `code` back
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
std::string& error_text();

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// ---- RCCL, loaded on first use so that single-GPU users never touch it -------------------
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                              hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
// the library is looked up once, by whichever thread asks first; nullptr if it cannot be loaded
Rccl* rccl();

// one halo segment: contiguous planes sent to / received from the lower and upper neighbour
struct Seg {
    float *send_lo, *recv_lo, *send_hi, *recv_hi;
    size_t n;   // floats per direction
};

}  // namespace fsi

#define FS_HIP(call)                                                                            \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fsi::fail(FS_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                             __FILE__, __LINE__);                                               \
    } while (0)

#define FS_NCCL(call)                                                                               \
    do {                                                                                            \
        ncclResult_t e_ = (call);                                                                   \
        if (e_ != ncclSuccess)                                                                      \
            return fsi::fail(FS_ERR_COMM, "%s failed: %s (%s:%d)", #call,                           \
                             fsi::rccl()->GetErrorString(e_), __FILE__, __LINE__);                  \
    } while (0)

struct fs_solver {
    fs_params prm{};
    FsGrid g{};
    FsFields f{};
    int device = 0;
    hipStream_t st = nullptr;         // compute stream
    float* pool = nullptr;            // one allocation holding all 13 fields
    float* vort[3] = {};              // dense vorticity arrays (one allocation), FS_FLAG_VORT_CONFINE only
    float* vort_lattice = nullptr;    // |omega| at the points the confined faces share
    // MGPCG work arrays (one allocation): r, z over all levels, p, Ap, x on the fine grid
    float *cg_r = nullptr, *cg_z = nullptr, *cg_zalt = nullptr, *cg_p = nullptr, *cg_Ap = nullptr, *cg_x = nullptr;
    float* cg_p2 = nullptr;           // single GPU: the second p array of the fused p-update / operator pass
    double *cg_partial = nullptr, *cg_out = nullptr;
    double* h_dot = nullptr;          // pinned
    // MGPCG on a z-slab: this rank's window of every level (owned planes plus G ghost planes per
    // non-wall side) inside the dense work arrays; vz = ghost planes of z that are exact right now
    struct MgWin {
        int nx, ny, nz;    // global dims of the level
        int z0, nzl;       // owned planes
        int G;             // ghost depth of the level (the same on every rank)
        int lo, hi;        // ghost planes held below / above: G, or 0 at a wall
        size_t off;        // offset of the window inside r and z
        int vz;
        int tb;            // the blocked smoother takes this level: -1 not tried yet, 0 no, 1 yes
        size_t cell() const { return (size_t)nx * ny; }
        int held() const { return lo + nzl + hi; }
    };
    std::vector<MgWin> mgw;
    bool cg_x_is_pressure = false;    // the reference's member x has become the pressure buffer
    float cg_rtr = 0.f;
    size_t field_elems = 0;           // floats per field including slack and ghost planes
    size_t origin = 0;                // offset of element (0,0,z0) inside a field
    float* host[FS_NUM_FIELDS] = {};  // pinned dense host copies, allocated on first use
    // asynchronous read-back: device snapshot of one field, copy stream, events (created by the
    // first fs_field_host_begin: a solver that never reads back asynchronously holds no third stream)
    float* snap = nullptr;
    hipStream_t copy_st = nullptr;
    hipEvent_t ev_snap = nullptr, ev_copied = nullptr;
    int snap_field = -1;              // field whose copy is in flight, -1 = none
    double* d_partial = nullptr;
    double* d_res = nullptr;
    int* d_flag = nullptr;
    double* h_res = nullptr;          // pinned
    int* h_flag = nullptr;            // pinned
    int frame = 0;
    int last_sweeps = 0;
    bool prev_held = false;           // np holds the iterate before the last sweep
    // slab decomposition
    int rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    fs_exchange_fn xfn = nullptr;     // host-staged transport (tests, fallback)
    void* xctx = nullptr;
    hipStream_t cs = nullptr;         // communication stream
    hipEvent_t ev_ready = nullptr, ev_done = nullptr;
    bool self_peer = false;           // FS_FLAG_TEST_SELF_PEER: both neighbours are this rank itself (tests)
    int xchg_streams = -1;            // FS_XCHG_STREAMS: 1 / 0 force the two-stream / one-stream ordering
    char* h_stage[4] = {};            // pinned: send_lo, recv_lo, send_hi, recv_hi
    size_t h_stage_cap = 0;
    int vel_ghost = 0;                // valid ghost planes of the velocity on each side
    bool scal_ghost_ok = false;
    bool div_fresh = false;           // div holds the divergence of the current velocity
    int p_ghost = 0;                  // valid ghost planes of p on each side
    bool halo_pending = false;        // an exchange is in flight; exchange_wait() orders the compute stream after it
    // FS_DEBUG_XCHG=1: every operation the exchange code enqueues is followed by a labelled event
    // ("mark") on its queue; a wait that exceeds the time limit reports, per queue, the last mark
    // that completed and the first that did not, i.e. which operation heads the stuck queue
    struct XMark { hipEvent_t ev; const char* what; int xchg; char queue; };
    int dbg_xchg = 0;
    int xchg_seq = 0;                 // exchanges begun so far
    // With several processes on one GPU (the host-staged test rig) a compute queue parked in
    // hipStreamWaitEvent(ev_done) was seen to keep the communication queue's next dispatch - the
    // very operation it waits for - from starting.  host_wait = the host waits instead (default of
    // the host-staged transport; FS_XCHG_HOST_WAIT=0 / 1 forces either on either transport).
    bool host_wait = false;
    int xchg_timeout_ms = 60000;      // FS_XCHG_TIMEOUT_MS: limit of a host-side wait inside an exchange
    std::vector<XMark> marks;         // oldest first; completed marks are recycled from the front
    // how long this rank's host and queues waited for halos (fs_halo_wait): host time inside
    // exchange calls that block, and, when stage timing is on, device time between the moment the
    // compute stream asks for the received planes and the moment they are there
    double xchg_host_wait_ms = 0.0;
    long long xchg_count = 0;
    struct HaloWait { hipEvent_t asked, got; };
    std::vector<HaloWait> halo_waits;        // event pairs on the compute stream around ev_done waits
    std::vector<hipEvent_t> halo_ev_free;
    double xchg_dev_wait_ms = 0.0;
    // captured steps (one GPU): a step only launches kernels whose arguments are the field pointers
    // and constants, so a graph captured for one assignment of the 13 buffers to their roles can be
    // replayed whenever the buffers are in that assignment again (the ping-pong swaps give the
    // assignment a period of two steps)
    struct HostState {
        FsFields f;
        bool prev_held, scal_ghost_ok, div_fresh;
        int vel_ghost, p_ghost;
    };
    struct StepGraph { HostState before, after; hipGraphExec_t exec; };
    std::vector<StepGraph> graphs;
    // stage timing with HIP events on the solver's own stream (fs_profile_*)
    bool profiling = false;
    std::vector<hipEvent_t> ev_free;
    struct Pending { int stage; hipEvent_t a, b; };
    std::vector<Pending> ev_pending;
    double prof_ms[FS_NUM_STAGES] = {};
    long long prof_launches[FS_NUM_STAGES] = {};
};

namespace fsi {

// Brackets one stage with a pair of events when profiling is on.
struct StageScope {
    fs_solver* s;
    int stage;
    hipEvent_t a = nullptr, b = nullptr;
    StageScope(fs_solver* s_, int stage_, int launches) : s(s_), stage(stage_)
    {
        if (!s->profiling) return;
        s->prof_launches[stage] += launches;
        for (hipEvent_t* e : {&a, &b}) {
            if (!s->ev_free.empty()) {
                *e = s->ev_free.back();
                s->ev_free.pop_back();
            } else if (hipEventCreate(e) != hipSuccess) {
                *e = nullptr;
            }
        }
        if (a && b) (void)hipEventRecord(a, s->st);
    }
    ~StageScope()
    {
        if (!s->profiling || !a || !b) return;
        (void)hipEventRecord(b, s->st);
        s->ev_pending.push_back({stage, a, b});
    }
};

// reads the halo flag of a slab back (synchronises): FS_ERR_HALO if a back-trace left the ghost planes
int check_flag(fs_solver* s);

// ---- fs_exchange.hip -------------------------------------------------------------------------
// FS_DEBUG_XCHG=2: a line on stderr before every call of the exchange code that could block the
// host, so the last line of a stuck rank names the call it sits in
#define FS_TRACE(s, what)                                                                      \
    do {                                                                                       \
        if ((s)->dbg_xchg >= 2) {                                                              \
            fprintf(stderr, "[fluidsolver rank %d] exchange %d: %s\n", (s)->rank, (s)->xchg_seq, what); \
            fflush(stderr);                                                                    \
        }                                                                                      \
    } while (0)

// Segment for `n` ghost planes of one array; base addresses local plane 0, `face` = the uz layout
Seg make_seg(const fs_solver* s, float* base, int n, bool face);
// FS_DEBUG_XCHG: label the point queue q has reached.  `what` must be a string literal.
void xmark(fs_solver* s, hipStream_t q, const char* what);
// hipStreamSynchronize with the time limit FS_XCHG_TIMEOUT_MS and a report of the queue heads
int bounded_sync(fs_solver* s, hipStream_t q, const char* where);
// Compute-stream work enqueued after exchange_begin() runs concurrently with the transfer until
// exchange_wait() makes the compute stream (or, with host_wait, the host) wait for the received planes.
int exchange_begin(fs_solver* s, const Seg* segs, int nseg);
int exchange_wait(fs_solver* s);
int exchange(fs_solver* s, const Seg* segs, int nseg);
int ensure_velocity_ghosts(fs_solver* s, int planes);
int ensure_scalar_ghosts(fs_solver* s);
int ensure_advect_ghosts(fs_solver* s);      // old velocity, T and rho in one group
int advect_ghosts_begin(fs_solver* s);       // the same exchange without waiting for it
int ensure_p_ghosts(fs_solver* s, int planes);
// adds the device-side halo waits that have completed to xchg_dev_wait_ms (synchronises the compute stream)
int drain_halo_waits(fs_solver* s);

// ---- fs_project_jacobi.hip ------------------------------------------------------------------------
int run_jacobi(fs_solver* s, int sweeps);
// sum r^2 over the whole grid (all ranks), binary64
int residual_sync(fs_solver* s, bool previous, double* out);

// ---- fs_project_mgpcg.hip -------------------------------------------------------------------------
int mg_alloc(fs_solver* s);        // work arrays of a single domain / of a slab's windows (zeroed on s->st)
int run_mgpcg(fs_solver* s);
int run_mgpcg_slab(fs_solver* s);
int mg_vcycle(fs_solver* s);       // z = M^-1 r on the solver's own arrays

// ---- fs_stages.hip / fs_step.hip --------------------------------------------------------------------
// fuse_div: let the scalar advection also write the divergence of the new velocity
int stage_advect(fs_solver* s, bool fuse_div);
int step_eager(fs_solver* s);

}  // namespace fsi
