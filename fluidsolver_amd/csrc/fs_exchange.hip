// fs_exchange.hip -- halo exchange of a z-slab and the validity bookkeeping of its ghost planes.
//
// The grid is cut into z-slabs, one per rank (one process per GPU).  A rank owns cell planes
// [z0, z1) and the uz face planes [z0, z1]; face plane z1 is computed by both neighbours from
// identical inputs, so results are bit-identical to one GPU.  Every array carries ghost planes on
// each z side, refreshed lazily, only when a stage needs them.  Planes are contiguous in memory, so
// a halo is one send and one receive per field and neighbour, straight from and into the arrays,
// with RCCL (ncclSend/ncclRecv in one group, on a communication stream ordered against the compute
// stream by two events per exchange) or, for tests and as a fallback, through pinned host buffers
// that a caller-supplied function moves.  The reference is single-GPU (CuSolver/Source/Solver.cu
// never calls cudaSetDevice and launches on the default stream only, e.g. :1153-1155): nothing
// here has a counterpart there.
#include "fs_solver_impl.h"

#include <dlfcn.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>

using namespace fsi;

namespace {

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

double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

namespace fsi {

Rccl* rccl()
{
    static Rccl r;
    static const bool ok = load_rccl(r);
    return ok ? &r : nullptr;
}

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

namespace {

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

}  // namespace

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

namespace {

// With stage timing on, a wait of the compute stream for halos is bracketed by two events on that
// stream: the time between them is how long the stream stood still for the transfer.
hipEvent_t halo_event(fs_solver* s)
{
    hipEvent_t e = nullptr;
    if (!s->halo_ev_free.empty()) {
        e = s->halo_ev_free.back();
        s->halo_ev_free.pop_back();
    } else if (hipEventCreate(&e) != hipSuccess) {
        e = nullptr;
    }
    return e;
}

struct HaloWaitScope {
    fs_solver* s;
    hipEvent_t a = nullptr, b = nullptr;
    explicit HaloWaitScope(fs_solver* s_) : s(s_)
    {
        if (!s || !s->profiling) return;
        a = halo_event(s);
        b = halo_event(s);
        if (a && b) (void)hipEventRecord(a, s->st);
    }
    ~HaloWaitScope()
    {
        if (!s) return;
        if (!a || !b) {
            if (a) s->halo_ev_free.push_back(a);
            if (b) s->halo_ev_free.push_back(b);
            return;
        }
        (void)hipEventRecord(b, s->st);
        s->halo_waits.push_back({a, b});
    }
};

}  // namespace

int drain_halo_waits(fs_solver* s)
{
    if (s->halo_waits.empty()) return FS_OK;
    FS_HIP(hipStreamSynchronize(s->st));
    for (const fs_solver::HaloWait& w : s->halo_waits) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, w.asked, w.got) == hipSuccess) s->xchg_dev_wait_ms += ms;
        s->halo_ev_free.push_back(w.asked);
        s->halo_ev_free.push_back(w.got);
    }
    s->halo_waits.clear();
    return FS_OK;
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
        const double t0 = now_ms();
        int rc = bounded_sync(s, s->cs, "host-side wait for the received planes");
        s->xchg_host_wait_ms += now_ms() - t0;
        if (rc) return rc;
        s->halo_pending = false;
        return FS_OK;
    }
    FS_TRACE(s, "hipStreamWaitEvent(compute, ev_done)");
    {
        HaloWaitScope hw(s);
        FS_HIP(hipStreamWaitEvent(s->st, s->ev_done, 0));
    }
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
    ++s->xchg_count;
    // one-stream ordering: the transfer itself is what the compute stream stands still for
    HaloWaitScope one_stream_wait(two_streams ? nullptr : s);
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
        const double t0 = now_ms();
        {
            int rc = bounded_sync(s, q, "halo exchange, device-to-host copies");
            if (rc) return rc;
        }
        FS_TRACE(s, "callback");
        int rc = s->xfn(s->xctx, FS_XCHG_HALO, lo ? s->h_stage[0] : nullptr, lo ? s->h_stage[1] : nullptr,
                        hi ? s->h_stage[2] : nullptr, hi ? s->h_stage[3] : nullptr, bytes);
        s->xchg_host_wait_ms += now_ms() - t0;
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

}  // namespace fsi
