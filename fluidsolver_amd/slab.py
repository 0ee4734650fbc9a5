"""z-slab decomposed solver: one process per GPU, launched by torch.distributed.run.

``SlabSolver`` wraps ``fs_create_slab`` of the C ABI.  torch.distributed is the control
plane only (hand the RCCL unique id to every rank, barriers, gathers for tests); the halo
exchange itself runs inside the library, on RCCL (``transport="rccl"``) or, for tests and as
a fallback, through host buffers moved by torch.distributed point-to-point calls
(``transport="host"``, works with the gloo backend and with ranks sharing one GPU).
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import FS_UZ, FsError, Solver, load_library

FS_UNIQUE_ID_BYTES = 128
FS_XCHG_HALO, FS_XCHG_SUM_F64 = 0, 1

_EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                           C.c_void_p, C.c_size_t)


def slab_range(nz, rank, world):
    """(z0, nz_local) of a rank: the first nz % world ranks get one plane more."""
    base, extra = divmod(nz, world)
    return rank * base + min(rank, extra), base + (1 if rank < extra else 0)


def _bind(L):
    if getattr(L, "_slab_bound", False):
        return
    L.fs_comm_unique_id.argtypes = [C.c_void_p]
    L.fs_create_slab.argtypes = [C.POINTER(_capi.Params), C.c_int, C.c_int, C.c_void_p, _EXCHANGE_FN,
                                 C.c_void_p, C.POINTER(C.c_void_p)]
    L.fs_slab_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 4
    L._slab_bound = True


class SlabSolver(Solver):
    """One rank's slab of a z-decomposed grid.  ``params`` describes the GLOBAL grid."""

    def __init__(self, params, rank, world, transport="rccl", group=None):
        super().__init__(params)
        _bind(self._L)
        self.rank, self.world = rank, world
        self.transport = transport
        self.group = group
        self._cb = None
        self.z0, self.nz_local = slab_range(params.nz, rank, world)

    # -- transports ----------------------------------------------------------------------
    def _unique_id(self):
        import torch
        import torch.distributed as dist
        buf = (C.c_char * FS_UNIQUE_ID_BYTES)()
        if self.rank == 0:
            self._check(self._L.fs_comm_unique_id(buf))
        t = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
        if dist.get_backend(self.group) == "nccl":
            t = t.cuda()
        dist.broadcast(t, src=0, group=self.group)
        raw = bytes(t.cpu().numpy().tobytes())
        return (C.c_char * FS_UNIQUE_ID_BYTES).from_buffer_copy(raw)

    def _host_callback(self):
        import torch
        import torch.distributed as dist
        rank, world, group = self.rank, self.world, self.group
        on_gpu = dist.get_backend(group) == "nccl"

        def as_tensor(ptr, nbytes):
            arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(nbytes,))
            return torch.from_numpy(arr)

        def fn(ctx, op, a, b, c, d, nbytes):
            try:
                if op == FS_XCHG_SUM_F64:
                    v = torch.from_numpy(np.ctypeslib.as_array(C.cast(a, C.POINTER(C.c_double)), shape=(1,)))
                    w = v.cuda() if on_gpu else v
                    dist.all_reduce(w, group=group)
                    if on_gpu:
                        v.copy_(w.cpu())
                    return 0
                reqs, pending = [], []
                for send, recv, peer in ((a, b, rank - 1), (c, d, rank + 1)):
                    if not send:
                        continue
                    st, rt = as_tensor(send, nbytes), as_tensor(recv, nbytes)
                    if on_gpu:
                        sg, rg = st.cuda(), torch.empty(nbytes, dtype=torch.uint8, device="cuda")
                        pending.append((rt, rg))
                        st, rt = sg, rg
                    reqs.append(dist.P2POp(dist.isend, st, peer, group))
                    reqs.append(dist.P2POp(dist.irecv, rt, peer, group))
                if reqs:
                    for w in dist.batch_isend_irecv(reqs):
                        w.wait()
                for host, dev in pending:
                    host.copy_(dev.cpu())
                return 0
            except Exception as e:  # never let an exception cross the C boundary
                import sys
                print("fluidsolver_amd.slab: exchange callback failed: %r" % (e,), file=sys.stderr)
                return 1

        return _EXCHANGE_FN(fn)

    # -- lifetime ------------------------------------------------------------------------
    def Initialize(self):
        if self._h is not None:
            raise FsError(-6, "Initialize called twice")
        h = C.c_void_p()
        if self.world == 1:
            self._check(self._L.fs_create(C.byref(self.params), C.byref(h)))
        elif self.transport == "rccl":
            uid = self._unique_id()
            self._check(self._L.fs_create_slab(C.byref(self.params), self.rank, self.world, uid,
                                               _EXCHANGE_FN(0), None, C.byref(h)))
        elif self.transport == "host":
            self._cb = self._host_callback()
            self._check(self._L.fs_create_slab(C.byref(self.params), self.rank, self.world, None,
                                               self._cb, None, C.byref(h)))
        else:
            raise ValueError("unknown transport %r" % (self.transport,))
        self._h = h
        return self

    # -- local / global field access ---------------------------------------------------------
    def local_shape(self, field):
        p = self.params
        nz = self.nz_local + (1 if field == FS_UZ else 0)
        return (nz,) + _capi.field_shape(field, p.nx, p.ny, p.nz)[1:]

    # field(), field_end(), velocity() and upload() of the base class size their arrays with
    # local_shape(), so on a slab they move the LOCAL planes

    def upload_global(self, field, arr):
        """Take this rank's planes out of a GLOBAL (z, y, x) array."""
        nz = self.nz_local + (1 if field == FS_UZ else 0)
        self.upload(field, np.asarray(arr)[self.z0:self.z0 + nz])

    def gather(self, field):
        """GLOBAL field on every rank (tests): owned planes of every rank, concatenated along z.
        For FS_UZ the shared face plane is taken from the lower rank."""
        import torch
        import torch.distributed as dist
        loc = self.field(field)
        if self.world == 1:
            return loc
        if field == FS_UZ and self.rank != self.world - 1:
            loc = loc[:-1]
        parts = [None] * self.world
        dist.all_gather_object(parts, loc, group=self.group)
        return np.concatenate(parts, axis=0)
