"""The path's one collective issued DIRECTLY on the launch stream: a ctypes binding of RCCL's ``ncclAllGather``.

``torch.distributed.all_gather_into_tensor`` runs the collective on ProcessGroupNCCL's own stream: an event recorded on the
launch stream, RCCL's stream waiting for it, and the launch stream waiting for RCCL's end event -- two cross-stream
dependencies, ~10 us of device idle per step on MI355X even where the collective itself is nothing (a one-rank group, in
place: ``bench.py`` under ``torch.distributed.run --nproc-per-node 1`` measured 0.2036 vs 0.1939 ms per step).  A voting
step is 60-200 us, so that is 5-15 % of it.  With the communicator in hand the same ``ncclAllGather`` goes onto the stream
the voting kernels were launched on: stream order is the only synchronisation, nothing is recorded or waited for, and the
whole step (vote + exchange) is capturable in one HIP graph.

``Comm`` is created once per process from an initialised ``torch.distributed`` group (any backend: only used to broadcast
the 128-byte ``ncclUniqueId`` and to agree on success); every rank must call it.  On any failure -- library not found,
init error, ranks sharing a GPU (RCCL refuses that) -- EVERY rank gets ``None`` and the caller keeps using torch's
collective: the decision is all-reduced, so the ranks cannot disagree.

The reference has no counterpart (no collective anywhere in it).
"""
import ctypes
import os

import torch
import torch.distributed as dist

_NCCL_FLOAT32 = 7          # ncclDataType_t: ncclFloat32 (nccl.h; RCCL keeps NCCL's enum)
_NCCL_SUCCESS = 0


class _UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_byte * 128)]


def _load():
    cands = [os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "librccl.so", "/opt/rocm/lib/librccl.so"]
    last = None
    for c in cands:                      # torch's own copy first: the process must not end up with two RCCL instances
        try:
            L = ctypes.CDLL(c)
            L.ncclGetErrorString.restype = ctypes.c_char_p
            L.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
            L.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _UniqueId, ctypes.c_int]
            L.ncclAllGather.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
            L.ncclCommDestroy.argtypes = [ctypes.c_void_p]
            return L
        except OSError as e:
            last = e
    raise OSError("librccl.so not found: %s" % (last,))


class Comm:
    """An RCCL communicator over the ranks of ``group`` (default: the world), one rank per GPU."""

    def __init__(self, lib, handle, world, rank):
        self._lib, self._h, self.world, self.rank = lib, handle, world, rank

    @staticmethod
    def create(device, group=None):
        """-> ``Comm`` on every rank, or ``None`` on every rank (see the module docstring).  Collective: call it on all ranks."""
        if not (dist.is_available() and dist.is_initialized()):
            return None
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        ctrl = torch.device("cpu") if dist.get_backend(group) == "gloo" else torch.device(device)
        ok, lib, handle, err = 1, None, ctypes.c_void_p(), ""
        uid = _UniqueId()
        try:
            lib = _load()
            if rank == 0:
                rc = lib.ncclGetUniqueId(ctypes.byref(uid))
                if rc != _NCCL_SUCCESS:
                    raise RuntimeError("ncclGetUniqueId: %s" % lib.ncclGetErrorString(rc).decode())
        except Exception as e:                                     # noqa: BLE001  (whatever it is: fall back, all ranks together)
            ok, err = 0, str(e)
        # every rank learns whether rank 0 has an id before anybody enters ncclCommInitRank (which would hang otherwise)
        flag = torch.tensor([ok], dtype=torch.int32, device=ctrl)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) == 0:
            return None
        t = torch.tensor(list(bytes(uid.internal)) if rank == 0 else [0] * 128, dtype=torch.uint8, device=ctrl)
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        ctypes.memmove(ctypes.byref(uid), bytes(t.cpu().tolist()), 128)
        try:
            with torch.cuda.device(device):
                rc = lib.ncclCommInitRank(ctypes.byref(handle), world, uid, rank)
            if rc != _NCCL_SUCCESS:
                raise RuntimeError("ncclCommInitRank: %s" % lib.ncclGetErrorString(rc).decode())
        except Exception as e:                                     # noqa: BLE001
            ok, err = 0, str(e)
        flag = torch.tensor([ok], dtype=torch.int32, device=ctrl)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) == 0:
            if ok and handle:
                lib.ncclCommDestroy(handle)
            Comm.last_error = err or "another rank failed to initialise its communicator"
            return None
        return Comm(lib, handle, world, rank)

    last_error = ""

    def all_gather_f32(self, send, recv, stream=None):
        """``recv[r * n : (r + 1) * n] = rank r's send`` for n = send.numel() float32 values, on ``stream`` (default: torch's current
        stream of the tensors' device).  In place when ``send`` is this rank's slice of ``recv``.  Stream-ordered: returns at once."""
        assert send.dtype == torch.float32 and recv.dtype == torch.float32 and send.is_contiguous() and recv.is_contiguous()
        assert recv.numel() == self.world * send.numel(), (recv.shape, send.shape, self.world)
        st = torch.cuda.current_stream(send.device).cuda_stream if stream is None else stream
        rc = self._lib.ncclAllGather(ctypes.c_void_p(send.data_ptr()), ctypes.c_void_p(recv.data_ptr()), send.numel(), _NCCL_FLOAT32,
                                     self._h, ctypes.c_void_p(st))
        if rc != _NCCL_SUCCESS:
            raise RuntimeError("ncclAllGather: %s" % self._lib.ncclGetErrorString(rc).decode())

    def destroy(self):
        if self._h:
            self._lib.ncclCommDestroy(self._h)
            self._h = ctypes.c_void_p()
