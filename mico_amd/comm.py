"""RCCL collectives through the C-ABI (`mico_comm_*` of include/mico_hip.h, csrc/comm.hip) - the transport a binding WITHOUT torch.distributed
would use, and an alternative one for mico_amd.distributed (MICO_COMM=1 or comm.enable()): the packed all-gather of the contrastive step as
pack kernel + ONE ncclAllGather on the compute stream, the index-then-fetch row exchange as a grouped send / receive, gradient averaging as an
in-place all-reduce.  Stream-ordered with the kernels around them (no side stream, no event hand-over).

Ordering against torch.distributed's own communicator: two communicators whose collectives can be in flight at the same time must be
issued in the same relative order on every rank.  With MICO_COMM=1 the collectives of THIS communicator are the forward's packed all-gather and row
exchange and the row exchange's mirrored return in the backward; GradBucketReducer's all-reduces (torch's communicator, its own stream) start inside the
final backward and are waited for by finish() before the next forward.  In the staged form of the step (MiCo.forward(backward_scale=...)) the return
exchange runs inside the forward, with the reducer's hooks off - the two communicators never overlap.  In the direct form the return exchange and the
first bucket reductions can overlap in the backward: every rank issues them from the same autograd order (same task, same graph), which is what RCCL
needs, but this has only run on a one-rank group (tests/test_distributed_gpu.py::test_mico_comm_c_abi_one_rank) - MICO_COMM stays opt-in until an
N > 1 run has exercised it.

The communicator's 128-byte id travels over whatever process group torch.distributed already has (any backend: it is a host-side broadcast);
a program without torch.distributed distributes it by its own means and calls MicoComm(rank, nranks, id) directly."""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import check

_ENABLED = os.environ.get("MICO_COMM", "0") == "1"
_COMM = None


def enable(on=True):
    """Route mico_amd.distributed's packed all-gather and row exchange through mico_comm_* (needs an initialised torch.distributed group to
    bootstrap the communicator, and CUDA tensors).  Returns the old setting."""
    global _ENABLED
    old, _ENABLED = _ENABLED, bool(on)
    return old


def enabled():
    return _ENABLED


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class MicoComm:
    def __init__(self, rank, nranks, uid):
        assert len(uid) == 128
        self.rank, self.nranks = int(rank), int(nranks)
        h = C.c_void_p()
        buf = (C.c_char * 128).from_buffer_copy(bytes(uid))
        check(_lib.lib().mico_comm_init(C.byref(h), self.rank, self.nranks, C.cast(buf, C.c_void_p)), "mico_comm_init")
        self._h = h

    @staticmethod
    def unique_id():
        buf = (C.c_char * 128)()
        check(_lib.lib().mico_comm_unique_id(C.cast(buf, C.c_void_p)), "mico_comm_unique_id")
        return bytes(buf)

    def close(self):
        if self._h is not None:
            torch.cuda.synchronize()
            check(_lib.lib().mico_comm_destroy(self._h), "mico_comm_destroy")
            self._h = None

    # ---- collectives (all on the current stream) ----
    def allgather_packed(self, tensors):
        """tensors: per-rank [b, ...] CUDA tensors of any dtypes -> list of [b * W, ...] tensors, ONE collective."""
        b = tensors[0].shape[0]
        flat = [t.detach().contiguous() for t in tensors]
        widths = [f.numel() // max(b, 1) * f.element_size() for f in flat]
        total = sum(widths)
        dev = flat[0].device
        scratch = torch.empty((b, total), dtype=torch.uint8, device=dev)
        out = torch.empty((self.nranks * b, total), dtype=torch.uint8, device=dev)
        parts = (C.c_void_p * len(flat))(*[f.data_ptr() for f in flat])
        rb = (C.c_int64 * len(flat))(*widths)
        check(_lib.lib().mico_comm_allgather_packed(self._h, parts, rb, len(flat), b, scratch.data_ptr(), out.data_ptr(), _st()),
              "mico_comm_allgather_packed")
        scratch.record_stream(torch.cuda.current_stream())
        res, o = [], 0
        for t, w in zip(tensors, widths):
            res.append(out[:, o:o + w].contiguous().view(t.dtype).view(self.nranks * b, *t.shape[1:]))
            o += w
        return res

    def alltoallv_rows(self, send, send_counts, recv_counts):
        """send [sum(send_counts), ...] split by destination rank -> [sum(recv_counts), ...] by source rank."""
        send = send.contiguous()
        row = send[0].numel() * send.element_size() if send.shape[0] else int(torch.tensor(send.shape[1:]).prod()) * send.element_size()
        recv = torch.empty((sum(recv_counts),) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        sb = (C.c_int64 * self.nranks)(*[c * row for c in send_counts])
        rb = (C.c_int64 * self.nranks)(*[c * row for c in recv_counts])
        check(_lib.lib().mico_comm_alltoallv(self._h, send.data_ptr() if send.numel() else None, sb, recv.data_ptr() if recv.numel() else None, rb, _st()),
              "mico_comm_alltoallv")
        return recv

    def allreduce_(self, flat, average=True):
        assert flat.dtype == torch.float32 and flat.is_contiguous()
        check(_lib.lib().mico_comm_allreduce_f32(self._h, flat.data_ptr(), flat.numel(), int(average), _st()), "mico_comm_allreduce_f32")
        return flat

    def reduce_scatter(self, flat, average=True):
        assert flat.dtype == torch.float32 and flat.is_contiguous() and flat.numel() % self.nranks == 0
        out = torch.empty(flat.numel() // self.nranks, dtype=torch.float32, device=flat.device)
        check(_lib.lib().mico_comm_reduce_scatter_f32(self._h, flat.data_ptr(), out.data_ptr(), out.numel(), int(average), _st()),
              "mico_comm_reduce_scatter_f32")
        return out

    def allgather(self, t):
        t = t.contiguous()
        out = torch.empty((self.nranks,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        check(_lib.lib().mico_comm_allgather(self._h, t.data_ptr(), out.data_ptr(), t.numel() * t.element_size(), _st()), "mico_comm_allgather")
        return out


def get():
    """The process's communicator over torch.distributed's world, created on first use (collective: every rank must get here)."""
    global _COMM
    import torch.distributed as dist
    if _COMM is None:
        rank, world = dist.get_rank(), dist.get_world_size()
        box = [MicoComm.unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        _COMM = MicoComm(rank, world, box[0])
    return _COMM


def shutdown():
    global _COMM
    if _COMM is not None:
        _COMM.close()
        _COMM = None
