// fs_solver.hip -- the solver object behind the C ABI of include/fluidsolver.h.
//
// Host-side orchestration of one simulation step on one z-slab: buffer ownership,
// stage order (reference: CuSolver/Source/Solver.cu:1009-1114 UpdateCuda, 1145-1165
// Advect, 1168-1189 Project on its Jacobi branch), ping-pong swaps (Solver.cu:181-186),
// host getters (Solver.cu:1368-1372).
//
// Multi-GPU: the grid is cut into z-slabs, one per rank (one process per GPU).  A rank owns
// cell planes [z0, z1) and the uz face planes [z0, z1]; face plane z1 is computed by both
// neighbours from identical inputs, so results are bit-identical to one GPU.  Every array
// carries `ghost` planes on each z side.  Ghost planes are refreshed lazily, only when a stage
// needs them: old velocity + T + rho before the velocity advection, the new velocity before the
// scalar advection.  A fused Jacobi pass first produces the jacobi_block() planes next to each
// neighbour, sends them (one exchange per 4 sweeps) and produces the slab interior while they
// travel; the gradient then finds the ghost planes of p already valid.  Planes are contiguous in
// memory, so a halo is one send and one receive per field and neighbour, straight from and into
// the arrays, with RCCL (ncclSend/ncclRecv in one group, on a communication stream ordered against
// the compute stream by events).  The only collective is the all-reduce of the residual norm.
#include "../../include/fluidsolver.h"
#include "fs_kernels.h"

#include <dlfcn.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <utility>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define FS_HIP(call)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(FS_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),   \
                        __FILE__, __LINE__);                                                 \
    } while (0)

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

bool load_rccl(Rccl& r)
{
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (r.lib) break;
    }
    if (!r.lib) return false;
#define FS_SYM(field, name)                                                   \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, name));        \
    if (!r.field) {                                                           \
        r.lib = nullptr;                                                      \
        return false;                                                         \
    }
    FS_SYM(GetUniqueId, "ncclGetUniqueId")
    FS_SYM(CommInitRank, "ncclCommInitRank")
    FS_SYM(CommDestroy, "ncclCommDestroy")
    FS_SYM(CommCount, "ncclCommCount")
    FS_SYM(Send, "ncclSend")
    FS_SYM(Recv, "ncclRecv")
    FS_SYM(GroupStart, "ncclGroupStart")
    FS_SYM(GroupEnd, "ncclGroupEnd")
    FS_SYM(AllReduce, "ncclAllReduce")
    FS_SYM(GetErrorString, "ncclGetErrorString")
#undef FS_SYM
    return true;
}

// the library is looked up once, by whichever thread asks first
Rccl* rccl()
{
    static Rccl r;
    static const bool ok = load_rccl(r);
    return ok ? &r : nullptr;
}

#define FS_NCCL(call)                                                                          \
    do {                                                                                       \
        ncclResult_t e_ = (call);                                                              \
        if (e_ != ncclSuccess)                                                                 \
            return fail(FS_ERR_COMM, "%s failed: %s (%s:%d)", #call, rccl()->GetErrorString(e_), \
                        __FILE__, __LINE__);                                                   \
    } while (0)

// one halo segment: contiguous planes sent to / received from the lower and upper neighbour
struct Seg {
    float *send_lo, *recv_lo, *send_hi, *recv_hi;
    size_t n;   // floats per direction
};

}  // namespace

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
    // asynchronous read-back: device snapshot of one field, copy stream, events
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
    // captured steps: a step only launches kernels whose arguments are the field pointers and
    // constants, so a graph captured for one assignment of the 13 buffers to their roles can be
    // replayed whenever the buffers are in that assignment again (the ping-pong swaps give the
    // assignment a period of two steps)
    // (a slab's graph also depends on which ghost planes were valid when it was captured: the
    // exchanges it contains are the ones that state asked for)
    struct HostState {
        FsFields f;
        bool prev_held, scal_ghost_ok, div_fresh;
        int vel_ghost, p_ghost;
    };
    struct StepGraph { HostState before, after; hipGraphExec_t exec; };
    std::vector<StepGraph> graphs;
    int eager_steps_left = 0;         // slabs: the first steps run eagerly (RCCL sets up its connections)
    bool graph_failed = false;        // a capture failed once: this solver launches eagerly from then on
    // stage timing with HIP events on the solver's own stream (fs_profile_*)
    bool profiling = false;
    std::vector<hipEvent_t> ev_free;
    struct Pending { int stage; hipEvent_t a, b; };
    std::vector<Pending> ev_pending;
    double prof_ms[FS_NUM_STAGES] = {};
    long long prof_launches[FS_NUM_STAGES] = {};
};

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

// ---- halo exchange -------------------------------------------------------------------------

// Segment for `n` ghost planes of one array.  base addresses local plane 0.  Cell-type arrays
// own planes [0, nzl); the face array uz owns [0, nzl] and shares plane nzl with the upper
// neighbour (its plane 0), so its exchanged ranges are shifted by one plane.
Seg make_seg(const fs_solver* s, float* base, int n, bool face)
{
    const size_t pl = (size_t)s->g.plane;
    const int nzl = s->g.nzl, f = face ? 1 : 0;
    Seg g;
    g.n = (size_t)n * pl;
    g.send_lo = s->g.wall_lo ? nullptr : base + (size_t)f * pl;
    g.recv_lo = s->g.wall_lo ? nullptr : base - (ptrdiff_t)n * (ptrdiff_t)pl;
    g.send_hi = s->g.wall_hi ? nullptr : base + (size_t)(nzl - n) * pl;
    g.recv_hi = s->g.wall_hi ? nullptr : base + (size_t)(nzl + f) * pl;
    return g;
}

// FS_DEBUG_XCHG=2: a line on stderr before every call of the exchange code that could block the
// host, so the last line of a stuck rank names the call it sits in
#define FS_TRACE(s, what)                                                                      \
    do {                                                                                       \
        if ((s)->dbg_xchg >= 2) {                                                              \
            fprintf(stderr, "[fluidsolver rank %d] exchange %d: %s\n", (s)->rank, (s)->xchg_seq, what); \
            fflush(stderr);                                                                    \
        }                                                                                      \
    } while (0)

// FS_DEBUG_XCHG: label the point both queues have reached.  `what` must be a string literal.
void xmark(fs_solver* s, hipStream_t q, const char* what)
{
    if (!s->dbg_xchg) return;
    hipEvent_t ev = nullptr;
    // recycle the oldest mark once it has completed and the list is long
    if (s->marks.size() >= 64 && hipEventQuery(s->marks.front().ev) == hipSuccess) {
        ev = s->marks.front().ev;
        s->marks.erase(s->marks.begin());
    } else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
        return;
    }
    if (hipEventRecord(ev, q) != hipSuccess) {
        (void)hipEventDestroy(ev);
        return;
    }
    s->marks.push_back({ev, what, s->xchg_seq, q == s->st ? 's' : 'c'});
}

const char* ready_str(hipError_t e)
{
    return e == hipSuccess ? "complete" : (e == hipErrorNotReady ? "NOT complete" : hipGetErrorString(e));
}

// What each queue is doing, for the report of a stalled wait.
void xchg_report(fs_solver* s, const char* where, double waited_s)
{
    fprintf(stderr, "[fluidsolver rank %d] %s: waited %.1f s in exchange %d; compute stream %s, communication "
                    "stream %s, ev_ready %s, ev_done %s\n", s->rank, where, waited_s, s->xchg_seq,
            ready_str(hipStreamQuery(s->st)), s->cs ? ready_str(hipStreamQuery(s->cs)) : "-",
            s->ev_ready ? ready_str(hipEventQuery(s->ev_ready)) : "-",
            s->ev_done ? ready_str(hipEventQuery(s->ev_done)) : "-");
    for (char queue : {'s', 'c'}) {
        const fs_solver::XMark *done = nullptr, *head = nullptr;
        int behind = 0;
        for (const fs_solver::XMark& m : s->marks) {
            if (m.queue != queue) continue;
            if (hipEventQuery(m.ev) == hipSuccess) {
                if (!head) done = &m;
            } else {
                if (!head) head = &m;
                else ++behind;
            }
        }
        if (!done && !head) continue;
        fprintf(stderr, "[fluidsolver rank %d]   %s queue: last completed '%s' (exchange %d); head of the queue "
                        "'%s' (exchange %d), %d marked operations behind it\n", s->rank,
                queue == 's' ? "compute" : "communication", done ? done->what : "-", done ? done->xchg : -1,
                head ? head->what : "(empty)", head ? head->xchg : -1, behind);
    }
    fflush(stderr);
}

// hipStreamSynchronize with a time limit: a stalled queue becomes an error with a report of what
// both queues hold instead of a silent hang.
int bounded_sync(fs_solver* s, hipStream_t q, const char* where)
{
    FS_TRACE(s, where);
    if (s->xchg_timeout_ms <= 0) {
        FS_HIP(hipStreamSynchronize(q));
        return FS_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    double next_report = s->dbg_xchg ? 2.0 : 1e30;
    for (long spins = 0;; ++spins) {
        hipError_t e = hipStreamQuery(q);
        if (e == hipSuccess) return FS_OK;
        if (e != hipErrorNotReady)
            return fail(FS_ERR_HIP, "%s: hipStreamQuery failed: %s", where, hipGetErrorString(e));
        if (spins < 2000) continue;
        usleep(spins < 20000 ? 20 : 200);
        const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (waited >= next_report) {
            xchg_report(s, where, waited);
            next_report *= 2.0;
        }
        if (waited * 1e3 >= s->xchg_timeout_ms) {
            xchg_report(s, where, waited);
            return fail(FS_ERR_COMM, "rank %d: %s did not finish within %d ms (FS_XCHG_TIMEOUT_MS); queue "
                                     "states are on stderr, FS_DEBUG_XCHG=1 adds the operation at the head of each queue",
                        s->rank, where, s->xchg_timeout_ms);
        }
    }
}

// Compute-stream work enqueued after exchange_begin() runs concurrently with the transfer (RCCL
// transport) until exchange_wait() makes the compute stream wait for the received planes.
int exchange_wait(fs_solver* s)
{
    if (!s->halo_pending) return FS_OK;
    if (s->host_wait) {
        // the host waits for the received planes instead of the compute queue, so no queue ever
        // parks in a wait on another queue (default of the host-staged transport, whose host is in
        // the middle of every exchange anyway; see FS_XCHG_HOST_WAIT)
        int rc = bounded_sync(s, s->cs, "host-side wait for the received planes");
        if (rc) return rc;
        s->halo_pending = false;
        return FS_OK;
    }
    FS_TRACE(s, "hipStreamWaitEvent(compute, ev_done)");
    FS_HIP(hipStreamWaitEvent(s->st, s->ev_done, 0));
    s->halo_pending = false;
    xmark(s, s->st, "wait for the received planes (ev_done)");
    return FS_OK;
}

// The transfer runs on the communication stream, ordered after everything enqueued on the compute
// stream so far (ev_ready), and overlaps with what the caller enqueues on the compute stream next;
// exchange_wait() orders the compute stream after it (ev_done).  Both transports share this
// ordering.  FS_XCHG_STREAMS=0 puts everything on the compute stream instead (no overlap).
int exchange_begin(fs_solver* s, const Seg* segs, int nseg)
{
    if (s->world == 1 || nseg == 0) return FS_OK;
    {
        int rc = exchange_wait(s);   // one exchange in flight at a time
        if (rc) return rc;
    }
    const bool lo = !s->g.wall_lo, hi = !s->g.wall_hi;
    const bool two_streams = s->xchg_streams != 0;
    hipStream_t q = s->st;
    ++s->xchg_seq;
    if (two_streams) {
        // the communication stream waits for the producers on the compute stream ...
        q = s->cs;
        FS_TRACE(s, "hipEventRecord(ev_ready, compute)");
        FS_HIP(hipEventRecord(s->ev_ready, s->st));
        xmark(s, s->st, "producers of the planes to send (ev_ready)");
        FS_TRACE(s, "hipStreamWaitEvent(communication, ev_ready)");
        FS_HIP(hipStreamWaitEvent(q, s->ev_ready, 0));
        xmark(s, q, "wait for the producers (ev_ready)");
    }
    if (s->comm) {
        Rccl* R = rccl();
        const int peer_lo = s->self_peer ? 0 : s->rank - 1, peer_hi = s->self_peer ? 0 : s->rank + 1;
        FS_NCCL(R->GroupStart());
        for (int i = 0; i < nseg; ++i) {
            const Seg& g = segs[i];
            if (lo) {
                FS_NCCL(R->Send(g.send_lo, g.n, ncclFloat, peer_lo, s->comm, q));
                FS_NCCL(R->Recv(g.recv_lo, g.n, ncclFloat, peer_lo, s->comm, q));
            }
            if (hi) {
                FS_NCCL(R->Send(g.send_hi, g.n, ncclFloat, peer_hi, s->comm, q));
                FS_NCCL(R->Recv(g.recv_hi, g.n, ncclFloat, peer_hi, s->comm, q));
            }
        }
        FS_NCCL(R->GroupEnd());
        xmark(s, q, "ncclSend/ncclRecv group");
    } else {
        // host-staged transport: pack through pinned buffers, let the caller's function move them
        size_t bytes = 0;
        for (int i = 0; i < nseg; ++i) bytes += segs[i].n * sizeof(float);
        if (bytes > s->h_stage_cap) {
            s->h_stage_cap = 0;   // stays 0 if an allocation below fails: the next call starts over
            for (char*& b : s->h_stage) {
                if (b) (void)hipHostFree(b);
                b = nullptr;
                // coherent (fine-grained) host memory: the GPU reads what the callback just wrote
                // and the host sees the packed planes once the kernel has completed
                FS_HIP(hipHostMalloc((void**)&b, bytes, hipHostMallocCoherent));
            }
            s->h_stage_cap = bytes;
        }
        // The planes are packed into / unpacked from the pinned buffers by a kernel of this library
        // that addresses the host memory directly, one launch per direction of the exchange.  (The
        // runtime's asynchronous copies were what stalled when this transport ran on the
        // communication stream with several ranks on one GPU: the host-to-device copies of the
        // received planes stayed at the head of that stream's queue, all of their predecessors
        // complete, with the copy engines enabled or disabled.)
        if (nseg > fsk::kCopyBatchMax / 2) return fail(FS_ERR_INVALID, "exchange of %d segments", nseg);
        fsk::CopyBatch out{}, in{};
        size_t off = 0;
        for (int i = 0; i < nseg; ++i) {
            const size_t nb = segs[i].n * sizeof(float);
            if (lo) {
                out.dst[out.count] = (float*)(s->h_stage[0] + off); out.src[out.count] = segs[i].send_lo; out.n[out.count++] = segs[i].n;
                in.dst[in.count] = segs[i].recv_lo; in.src[in.count] = (const float*)(s->h_stage[1] + off); in.n[in.count++] = segs[i].n;
            }
            if (hi) {
                out.dst[out.count] = (float*)(s->h_stage[2] + off); out.src[out.count] = segs[i].send_hi; out.n[out.count++] = segs[i].n;
                in.dst[in.count] = segs[i].recv_hi; in.src[in.count] = (const float*)(s->h_stage[3] + off); in.n[in.count++] = segs[i].n;
            }
            off += nb;
        }
        FS_TRACE(s, "launch pack kernel");
        fsk::copy_batch(q, out);
        FS_HIP(hipGetLastError());
        xmark(s, q, "copies of the planes to send into pinned memory");
        {
            int rc = bounded_sync(s, q, "halo exchange, device-to-host copies");
            if (rc) return rc;
        }
        FS_TRACE(s, "callback");
        int rc = s->xfn(s->xctx, FS_XCHG_HALO, lo ? s->h_stage[0] : nullptr, lo ? s->h_stage[1] : nullptr,
                        hi ? s->h_stage[2] : nullptr, hi ? s->h_stage[3] : nullptr, bytes);
        if (rc) return fail(FS_ERR_COMM, "rank %d: the exchange callback returned %d", s->rank, rc);
        FS_TRACE(s, "launch unpack kernel");
        fsk::copy_batch(q, in);
        FS_HIP(hipGetLastError());
        xmark(s, q, "copies of the received planes into the ghost planes");
    }
    if (two_streams) {
        // ... and the consumers on the compute stream wait for the halos in exchange_wait()
        FS_TRACE(s, "hipEventRecord(ev_done, communication)");
        FS_HIP(hipEventRecord(s->ev_done, q));
        s->halo_pending = true;
    }
    FS_TRACE(s, "exchange_begin returns");
    return FS_OK;
}

int exchange(fs_solver* s, const Seg* segs, int nseg)
{
    int rc = exchange_begin(s, segs, nseg);
    return rc ? rc : exchange_wait(s);
}

int ensure_velocity_ghosts(fs_solver* s, int planes)
{
    if (s->world == 1 || s->vel_ghost >= planes) return FS_OK;
    Seg segs[3] = {make_seg(s, s->f.ux, planes, false), make_seg(s, s->f.uy, planes, false),
                   make_seg(s, s->f.uz, planes, true)};
    int rc = exchange(s, segs, 3);
    if (rc == FS_OK) s->vel_ghost = planes;
    return rc;
}

int ensure_scalar_ghosts(fs_solver* s)
{
    if (s->world == 1 || s->scal_ghost_ok) return FS_OK;
    Seg segs[2] = {make_seg(s, s->f.T, s->g.ghost, false), make_seg(s, s->f.rho, s->g.ghost, false)};
    int rc = exchange(s, segs, 2);
    if (rc == FS_OK) s->scal_ghost_ok = true;
    return rc;
}

// old velocity, T and rho in one group (what the advection of a step starts from)
int ensure_advect_ghosts(fs_solver* s)
{
    if (s->world == 1) return FS_OK;
    Seg segs[5];
    int n = 0;
    const bool vel = s->vel_ghost < s->g.ghost;
    if (vel) {
        segs[n++] = make_seg(s, s->f.ux, s->g.ghost, false);
        segs[n++] = make_seg(s, s->f.uy, s->g.ghost, false);
        segs[n++] = make_seg(s, s->f.uz, s->g.ghost, true);
    }
    if (!s->scal_ghost_ok) {
        segs[n++] = make_seg(s, s->f.T, s->g.ghost, false);
        segs[n++] = make_seg(s, s->f.rho, s->g.ghost, false);
    }
    int rc = exchange(s, segs, n);
    if (rc == FS_OK) {
        if (vel) s->vel_ghost = s->g.ghost;
        s->scal_ghost_ok = true;
    }
    return rc;
}

// start the exchange ensure_advect_ghosts would make, without waiting for it; the caller sets the
// validity flags after exchange_wait()
int advect_ghosts_begin(fs_solver* s)
{
    Seg segs[5];
    int n = 0;
    if (s->vel_ghost < s->g.ghost) {
        segs[n++] = make_seg(s, s->f.ux, s->g.ghost, false);
        segs[n++] = make_seg(s, s->f.uy, s->g.ghost, false);
        segs[n++] = make_seg(s, s->f.uz, s->g.ghost, true);
    }
    if (!s->scal_ghost_ok) {
        segs[n++] = make_seg(s, s->f.T, s->g.ghost, false);
        segs[n++] = make_seg(s, s->f.rho, s->g.ghost, false);
    }
    return exchange_begin(s, segs, n);
}

int ensure_p_ghosts(fs_solver* s, int planes)
{
    if (s->world == 1) return FS_OK;
    if (s->p_ghost >= planes) return exchange_wait(s);
    Seg seg = make_seg(s, s->f.p, planes, false);
    int rc = exchange(s, &seg, 1);
    if (rc == FS_OK) s->p_ghost = planes;
    return rc;
}

// ---- Jacobi ----------------------------------------------------------------------------------

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

// ---- MGPCG -------------------------------------------------------------------------------------

// the sum a fused kernel pair has left in cg_out, as the float the host logic reads (synchronises)
int cg_read_sum(fs_solver* s, float* out)
{
    FS_HIP(hipMemcpyAsync(s->h_dot, s->cg_out, sizeof(double), hipMemcpyDeviceToHost, s->st));
    FS_HIP(hipStreamSynchronize(s->st));
    *out = (float)*s->h_dot;
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
    const int kpar = (W.z0 - W.lo) & 1, phases = 2 * steps;
    // (the blocked kernel takes the colour of a cell from the window's own plane numbers)
    bool blocked = kpar == 0 && W.G >= 2 && W.tb != 0;
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
        if (blocked && fsk::mgs_smooth_pass(s->st, r, z, zalt, W.nx, W.ny, W.held(), take, zero_in)) {
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
    const int ghost = world > 1 ? 4 : 2;
    int pghost = ghost;
    if (world > 1) {
        const char* e = getenv("FS_SLAB_PGHOST");
        const int want = e ? atoi(e) : 8;
        if (want >= 8 && p->nz / world >= 9) pghost = 8;
    }
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
    // so that the entry point itself never has to
    {
        hipError_t e = fsk::residual_static_lookup(dev);
        if (e != hipSuccess)
            return fail(FS_ERR_HIP, "fs_create: hipGetSymbolAddress of the static scratch on device %d: %s", dev,
                        hipGetErrorString(e));
    }

    fs_solver* s = new (std::nothrow) fs_solver;
    if (!s) return fail(FS_ERR_INVALID, "fs_create: out of host memory");
    s->prm = *p;
    s->device = dev;
    s->rank = rank;
    s->world = world;
    s->xfn = fn;
    s->xctx = ctx;
    FsGrid& g = s->g;
    g.nx = p->nx;
    g.ny = p->ny;
    g.nz = p->nz;
    g.z0 = z0;
    g.nzl = nzl;
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
        if (p->pressure_solver == FS_PRESSURE_MGPCG && world > 1) {
            // windows of the levels: owned planes plus G ghost planes per non-wall side.  G planes
            // last G colours of smoothing between two exchanges, so the coarse levels, whose planes
            // are small and whose sweeps are many, get deeper ghosts than the fine ones (a level
            // with large planes: 4, what one blocked pass uses up); never deeper than the slab of
            // the level, nor than twice the next coarser level's (prolongation halves the depth).
            // FS_MG_SLAB_GHOST caps every level (tests).
            const int L = p->mg_level;
            int cap_all = 1 << 20;
            if (const char* e = getenv("FS_MG_SLAB_GHOST"))
                if (atoi(e) >= 1) cap_all = atoi(e);
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
            const size_t space_al = off, n = (s->mgw[0].cell() * s->mgw[0].held() + 63) & ~(size_t)63;
            const size_t total = (3 * space_al + 3 * n) * sizeof(float);
            e = hipMalloc((void**)&s->cg_r, total);
            if (e != hipSuccess || hipMemsetAsync(s->cg_r, 0, total, s->st) != hipSuccess ||
                hipMalloc((void**)&s->cg_partial, (fsk::cg_dot_partials() + 4) * sizeof(double)) != hipSuccess ||
                hipHostMalloc((void**)&s->h_dot, sizeof(double)) != hipSuccess) {
                rc = fail(FS_ERR_HIP, "fs_create_slab: MGPCG work arrays (%zu bytes)", total);
                break;
            }
            s->cg_z = s->cg_r + space_al;
            s->cg_zalt = s->cg_z + space_al;
            s->cg_p = s->cg_zalt + space_al;
            s->cg_Ap = s->cg_p + n;
            s->cg_x = s->cg_Ap + n;
            s->cg_out = s->cg_partial + fsk::cg_dot_partials();
        } else if (p->pressure_solver == FS_PRESSURE_MGPCG) {
            const size_t space = fsk::mg_space(g, p->mg_level), n = (size_t)g.nx * g.ny * g.nz;
            const size_t space_al = (space + 63) & ~(size_t)63;   // keeps every array 256-byte aligned
            const size_t total = (3 * space_al + 4 * n) * sizeof(float);
            e = hipMalloc((void**)&s->cg_r, total);
            if (e != hipSuccess || hipMemsetAsync(s->cg_r, 0, total, s->st) != hipSuccess ||
                hipMalloc((void**)&s->cg_partial, (fsk::cg_dot_partials() + 4) * sizeof(double)) != hipSuccess ||
                hipHostMalloc((void**)&s->h_dot, sizeof(double)) != hipSuccess) {
                rc = fail(FS_ERR_HIP, "fs_create: MGPCG work arrays (%zu bytes)", total);
                break;
            }
            s->cg_z = s->cg_r + space_al;
            s->cg_zalt = s->cg_z + space_al;
            s->cg_p = s->cg_zalt + space_al;
            s->cg_Ap = s->cg_p + n;
            s->cg_x = s->cg_Ap + n;
            s->cg_p2 = s->cg_x + n;
            s->cg_out = s->cg_partial + fsk::cg_dot_partials();
        }
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
            const char* le = getenv("FS_VORT_LATTICE");
            if (!(le && atoi(le) == 0) &&
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
            if (const char* e = getenv("FS_XCHG_STREAMS")) s->xchg_streams = atoi(e) != 0;
            if (const char* e = getenv("FS_DEBUG_XCHG")) s->dbg_xchg = atoi(e);
            s->host_wait = unique_id == nullptr;
            if (const char* e = getenv("FS_XCHG_HOST_WAIT")) s->host_wait = atoi(e) != 0;
            if (const char* e = getenv("FS_XCHG_TIMEOUT_MS")) s->xchg_timeout_ms = atoi(e);
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
        fsk::jacobi_autotune(s->st, g, s->f.p, s->f.np, s->f.div);
        // zero state: every ghost plane already equals its neighbour's owned plane
        s->vel_ghost = g.pghost;
        s->scal_ghost_ok = true;
        s->p_ghost = g.pghost;
        s->eager_steps_left = world > 1 ? 2 : 0;
        if (hipStreamSynchronize(s->st) != hipSuccess || hipGetLastError() != hipSuccess) {
            rc = fail(FS_ERR_HIP, "fs_create: initialisation kernels failed");
            break;
        }
    } while (0);
    if (rc != FS_OK) {
        std::string keep = g_err;
        fs_destroy(s);
        g_err = keep;
        return rc;
    }
    *out = s;
    return FS_OK;
}

}  // namespace

extern "C" {

const char* fs_last_error(void) { return g_err.c_str(); }

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

// fuse_div: let the scalar advection also write the divergence of the new velocity (only inside
// fs_step, where the projection follows at once; a stand-alone fs_stage_advect leaves div alone,
// as Solver::Advect does)
static int stage_advect(fs_solver* s, bool fuse_div)
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

static int step_eager(fs_solver* s)
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

static fs_solver::HostState host_state(const fs_solver* s)
{
    return {s->f, s->prev_held, s->scal_ghost_ok, s->div_fresh, s->vel_ghost, s->p_ghost};
}

static void set_host_state(fs_solver* s, const fs_solver::HostState& h)
{
    s->f = h.f;
    s->prev_held = h.prev_held;
    s->vel_ghost = h.vel_ghost;
    s->scal_ghost_ok = h.scal_ghost_ok;
    s->div_fresh = h.div_fresh;
    s->p_ghost = h.p_ghost;
}

// what decides which launches and exchanges a step makes: the buffer assignment and, on a slab,
// the validity of the ghost planes
static bool same_launches(const fs_solver* s, const fs_solver::HostState& a, const fs_solver::HostState& b)
{
    if (std::memcmp(&a.f, &b.f, sizeof(FsFields)) != 0) return false;
    if (s->world == 1) return true;
    return a.vel_ghost == b.vel_ghost && a.scal_ghost_ok == b.scal_ghost_ok && a.p_ghost == b.p_ghost &&
           a.div_fresh == b.div_fresh;
}

// One step as a graph replay: fixed sweep count, no stage timing; one GPU (a slab on the RCCL
// transport only with FS_SLAB_GRAPH=1, see fs_step: its sends and receives would be captured on
// the communication stream, which the event pair of each exchange forks from and joins back into
// the compute stream).
static int step_graph(fs_solver* s)
{
    FS_HIP(hipSetDevice(s->device));
    const fs_solver::HostState now = host_state(s);
    for (const fs_solver::StepGraph& g : s->graphs) {
        if (!same_launches(s, g.before, now)) continue;
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
        s->halo_pending = false;
        if (s->world == 1) {
            if (rc) return rc;
            return fail(FS_ERR_HIP, "capture of a step failed: %s", hipGetErrorString(e));
        }
        // a slab falls back to eager launches for good (a communication library that cannot be
        // captured on this machine must not cost the run)
        s->graph_failed = true;
        return step_eager(s);
    }
    g.after = host_state(s);
    s->graphs.push_back(g);
    FS_HIP(hipGraphLaunch(g.exec, s->st));
    return FS_OK;
}

int fs_step(fs_solver* s)
{
    if (!s) return fail(FS_ERR_INVALID, "null solver");
    // (the CG loop reads its scalars on the host, so an MGPCG step cannot be captured)
    bool graph = s->prm.tolerance <= 0.f && !s->profiling && s->prm.pressure_solver == FS_PRESSURE_JACOBI &&
                 !(s->prm.flags & FS_FLAG_NO_GRAPH) && !s->graph_failed && !s->dbg_xchg && !s->host_wait;
    if (s->world > 1) {
        // Slabs launch eagerly.  Capturing a slab step (RCCL transport only, after a few eager
        // steps in which RCCL sets up its connections) is wired up behind FS_SLAB_GRAPH=1 but stays
        // off: with RCCL 2.26.6 the captured ncclSend/ncclRecv group of a rank that is its own
        // neighbour (the only form a one-GPU machine can run) ended in a segmentation fault inside
        // the library, and the capture has never run between two devices.
        static const bool slab_graph = [] {
            const char* e = getenv("FS_SLAB_GRAPH");
            return e && atoi(e) != 0;
        }();
        graph = graph && slab_graph && s->comm != nullptr;
        if (graph && s->eager_steps_left > 0) {
            --s->eager_steps_left;
            graph = false;
        }
    }
    int rc = graph ? step_graph(s) : step_eager(s);
    if (rc) return rc;
    s->frame++;
    return FS_OK;
}

int fs_step_graphs(const fs_solver* s)
{
    if (!s) return -2;
    return s->graph_failed ? -1 : (int)s->graphs.size();
}

int fs_steps(fs_solver* s, int n)
{
    for (int q = 0; q < n; ++q) {
        int rc = fs_step(s);
        if (rc) return rc;
    }
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

int fs_profile_reset(fs_solver* s)
{
    if (!s) return fail(FS_ERR_INVALID, "null solver");
    int rc = drain_events(s);
    if (rc) return rc;
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
