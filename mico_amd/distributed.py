"""Data-parallel exchange steps of the alignment loss, one process per GPU over torch.distributed (backend "nccl" is RCCL
over xGMI on ROCm; "gloo" on CPU for the tests).  Mirrors data/utils/distributed.py of the reference:

  concat_all_gather     (:50-66)  no-grad all-gather + cat - kept;
  all_gather_with_grad  (:12-47)  kept for API parity, but MiCo.forward does NOT use it for condition_feats: the reference
                                  gathers the whole [b*W, E, 768] memory (2-5.7 GB at W=8) and then reads only b rows of it
                                  (vast.py:430-433).  fetch_rows() exchanges just those rows (index-then-fetch) with the
                                  mirrored gradient return - identical values and gradients, ~W x less xGMI traffic;
  packed_all_gather     new       one collective for all per-step small tensors (features, ids, masks) instead of 4-6
                                  latency-bound launches.

Every function degrades to the identity when torch.distributed is not initialised (W = 1) - unless `force_dist(True)` /
MICO_FORCE_DIST=1 is set: then an initialised process group of ONE rank takes the N > 1 code paths too (the collectives, the
row exchange, the in-place arena-slice reduction), so that a 1-GPU box can run RCCL's `all_gather_into_tensor` /
`all_to_all_single` / `all_reduce(AVG)` calls exactly as an 8-GPU job issues them (tests/test_distributed_gpu.py, bench.py `comm`).
"""
import os

import torch
import torch.distributed as dist

_FORCE = os.environ.get("MICO_FORCE_DIST", "0") == "1"


def force_dist(on=True):
    """Take the distributed code paths at world size 1 as well (needs an initialised process group).  Returns the old setting."""
    global _FORCE
    old, _FORCE = _FORCE, bool(on)
    return old


_LOCAL = [0]


class local_only:
    """Context in which an initialised process group is ignored (is_dist() is False): work ONE rank does on its own while the others wait - e.g.
    bench.py's in-run parity measurement on rank 0 of an N > 1 job, whose alignment step would otherwise enter collectives nobody else joins."""

    def __enter__(self):
        _LOCAL[0] += 1

    def __exit__(self, *exc):
        _LOCAL[0] -= 1
        return False


def _nccl():
    return dist.get_backend() == "nccl"


def is_dist():
    return not _LOCAL[0] and dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _FORCE)


def world_size():
    return dist.get_world_size() if is_dist() else 1


def rank():
    return dist.get_rank() if is_dist() else 0


@torch.no_grad()
def concat_all_gather(tensor):
    if not is_dist():
        return tensor.detach()
    out = [torch.empty_like(tensor) for _ in range(dist.get_world_size())]
    dist.all_gather(out, tensor.contiguous())
    return torch.cat(out, dim=0)


class _GatherLayer(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        out = [torch.zeros_like(x) for _ in range(dist.get_world_size())]
        dist.all_gather(out, x.contiguous())
        return tuple(out)

    @staticmethod
    def backward(ctx, *grads):
        g = torch.stack(grads)
        dist.all_reduce(g)
        return g[dist.get_rank()]


def all_gather_with_grad(tensor):
    if not is_dist():
        return tensor
    return torch.cat(_GatherLayer.apply(tensor), dim=0)


@torch.no_grad()
def packed_all_gather(tensors):
    """All-gathers a list of per-rank tensors (same leading dim b, any dtypes) with ONE collective: everything is bit-cast
    into a single uint8 buffer [b, bytes], gathered, and unpacked to [b*W, ...] tensors (constants for autograd, exactly
    like concat_all_gather)."""
    if not is_dist():
        return [t.detach() for t in tensors]
    from . import comm
    # MICO_COMM=1: pack kernel + one ncclAllGather on the compute stream (mico_comm_allgather_packed: at most 8 parts - a task with more than five
    # retrieval sub-tasks takes the torch.distributed path below instead of failing)
    if comm.enabled() and tensors[0].is_cuda and len(tensors) <= 8:
        return comm.get().allgather_packed(tensors)
    b = tensors[0].shape[0]
    flat = [t.detach().contiguous().view(b, -1).view(torch.uint8) for t in tensors]
    widths = [f.shape[1] for f in flat]
    buf = torch.cat(flat, dim=1).contiguous()
    W = dist.get_world_size()
    out = torch.empty((W * b, buf.shape[1]), dtype=torch.uint8, device=buf.device)
    if _nccl():
        dist.all_gather_into_tensor(out, buf)
    else:
        dist.all_gather(list(out.chunk(W, 0)), buf)
    res, o = [], 0
    for t, w in zip(tensors, widths):
        piece = out[:, o:o + w].contiguous().view(t.dtype).view(W * b, *t.shape[1:])
        res.append(piece)
        o += w
    return res


def _exchange(send, send_counts, recv_counts):
    """Variable-size row exchange: send[sum(send_counts), ...] split by destination rank -> received rows by source rank."""
    W = dist.get_world_size()
    from . import comm
    if comm.enabled() and send.is_cuda:            # MICO_COMM=1: grouped ncclSend / ncclRecv on the compute stream (mico_comm_alltoallv)
        return comm.get().alltoallv_rows(send, send_counts, recv_counts)
    recv = torch.empty((sum(recv_counts),) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
    if _nccl():
        dist.all_to_all_single(recv, send.contiguous(), output_split_sizes=recv_counts, input_split_sizes=send_counts)
    else:   # gloo (tests): emulate with a padded all_gather
        mx = torch.tensor([max(send_counts + [1])], device=send.device)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        m = int(mx.item())
        pad = torch.zeros((W, m) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        o = 0
        for d, c in enumerate(send_counts):
            pad[d, :c] = send[o:o + c]
            o += c
        allp = [torch.empty_like(pad) for _ in range(W)]
        dist.all_gather(allp, pad)
        me = dist.get_rank()
        o = 0
        for s, c in enumerate(recv_counts):
            recv[o:o + c] = allp[s][me, :c]
            o += c
    return recv


class _FetchRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, local, global_idx):
        """local: this rank's [b, ...] rows (rank r owns global rows r*b .. r*b+b-1); global_idx [k] int64: rows wanted.
        Returns the [k, ...] requested rows.  One tiny index all-gather (+ host read of the counts) and one row exchange."""
        W, me = dist.get_world_size(), dist.get_rank()
        b = local.shape[0]
        k = global_idx.shape[0]
        idx_all = [torch.empty_like(global_idx) for _ in range(W)]
        dist.all_gather(idx_all, global_idx.contiguous())
        idx_all = torch.stack(idx_all).cpu()            # [W, k]   (single host sync; the reference does 2*b .item() syncs)
        owner = idx_all // b
        # what I must send: for requester q, the rows q wants from me, in q's request order
        send_ids, send_counts = [], []
        for q in range(W):
            sel = idx_all[q][owner[q] == me] - me * b
            send_ids.append(sel)
            send_counts.append(int(sel.numel()))
        send_ids = torch.cat(send_ids).to(local.device)
        # what I receive: my requests grouped by owner rank (order within an owner = my request order)
        my_owner = owner[me]
        recv_counts = [int((my_owner == s).sum()) for s in range(W)]
        order = torch.argsort(my_owner, stable=True)     # received position -> request slot
        recv = _exchange(local.detach().index_select(0, send_ids), send_counts, recv_counts)
        out = torch.empty_like(recv)
        out[order.to(local.device)] = recv
        ctx.save_for_backward(send_ids, order.to(local.device))
        ctx.counts = (send_counts, recv_counts, b)
        return out

    @staticmethod
    def backward(ctx, dout):
        send_ids, order = ctx.saved_tensors
        send_counts, recv_counts, b = ctx.counts
        back = _exchange(dout.contiguous()[order], recv_counts, send_counts)   # mirrored route
        dlocal = torch.zeros((b,) + tuple(dout.shape[1:]), dtype=dout.dtype, device=dout.device)
        dlocal.index_add_(0, send_ids, back)
        return dlocal, None


def fetch_rows(local, global_idx):
    """rows `global_idx` of the (virtual) concatenation of every rank's `local`, with gradients routed back to their
    owners - replaces all_gather_with_grad(condition_feats)[neg_idx] (vast.py:421-433)."""
    if not is_dist():
        return local.index_select(0, global_idx)
    return _FetchRows.apply(local, global_idx)


_STAGED = [0]


class staged_backward:
    """Context of a backward pass that runs INSIDE a forward (MiCo.forward(backward_scale=...): the BERT passes of one condition set): the
    parameters it reaches may be reached again by later staged passes and by the step's final backward, so GradBucketReducer's
    per-parameter hooks must not count these accumulations - the buckets of such parameters are reduced when the final backward completes them,
    or by finish()."""

    def __enter__(self):
        _STAGED[0] += 1

    def __exit__(self, *exc):
        _STAGED[0] -= 1
        return False


class GradBucketReducer:
    """Bucketed data-parallel gradient averaging overlapped with backward (the role DDP plays at
    data/utils/build_model.py:57).  Parameters are grouped in reverse registration order into flat fp32 buckets; when the
    last gradient of a bucket has been accumulated its all-reduce is launched asynchronously on RCCL's stream.  Bucket size
    defaults to 256 MiB: xGMI collectives are per-link bound, so few large messages beat many small ones."""

    def __init__(self, params, bucket_bytes=256 << 20):
        self.params = [p for p in params if p.requires_grad]
        self.buckets, cur, size = [], [], 0
        for p in reversed(self.params):
            cur.append(p)
            size += p.numel() * 4
            if size >= bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        self._pending = {}
        self._handles = []
        self._where = {}
        for bi, bucket in enumerate(self.buckets):
            for p in bucket:
                self._where[p] = bi
        self._early = set()          # ids of parameters whose gradient was already reduced inside a backward (arena slices)
        self._early_handles = []
        self._early_slices = []      # (flat arena slice, its parameters) of those reductions: finish() checks the aliasing
        self._hook_handles = []
        if is_dist():
            for p in self.params:
                self._hook_handles.append(p.register_post_accumulate_grad_hook(self._hook))
            from . import runtime
            runtime.set_grad_slice_hook(self._reduce_slice)
        self.reset()

    def close(self):
        """Detaches the reducer from the parameters and the tower backward (before its process group goes away)."""
        for h in self._hook_handles:
            h.remove()
        self._hook_handles = []
        from . import runtime
        runtime.set_grad_slice_hook(None)

    def _reduce_slice(self, flat, params):
        """runtime.grad_slice_hook: `flat` is the final gradient of `params` (one contiguous arena slice, e.g. a ViT block):
        all-reduce it in place right away - no bucket copy, and it overlaps with the backward of the earlier blocks."""
        if not is_dist():     # the process group is gone (a reducer outliving its group): nothing to reduce
            return
        # In-place reduction of the arena slice is only right when autograd will ADOPT the arena views as the parameters' .grad, i.e.
        # when none of them has a gradient yet (zero_grad(set_to_none=True), one backward per step).  With an existing .grad
        # (accumulation, set_to_none=False) autograd ADDS the view into it on the compute stream - while RCCL would be reducing that
        # memory - and the result would stay rank-local: leave such slices to the bucket path, which reduces the accumulated .grad.
        if any(p.grad is not None for p in params):
            return
        self._early_slices.append((flat, list(params)))
        if _nccl():
            h = dist.all_reduce(flat, op=dist.ReduceOp.AVG, async_op=True)
            self._early_handles.append((h, None, flat))
        else:
            h = dist.all_reduce(flat, async_op=True)
            self._early_handles.append((h, flat, flat))
        self._early.update(id(p) for p in params)

    def reset(self):
        self._pending = {bi: len(b) for bi, b in enumerate(self.buckets)}
        self._seen = {}
        self._launched = set()
        self._handles = []
        self._early = set()
        self._early_handles = []
        self._early_slices = []

    def _hook(self, p):
        if _STAGED[0]:      # a staged backward inside the forward: not the parameter's last accumulation of the step (see staged_backward)
            return
        bi = self._where[p]
        self._pending[bi] -= 1
        if self._seen.get(id(p), 0) > 0:
            raise RuntimeError("GradBucketReducer: a parameter's gradient was accumulated twice before finish() (two backward passes in one "
                               "step?) - its bucket was already reduced; call finish() after every backward, or accumulate locally and "
                               "build the reducer for the last micro-step only")
        self._seen[id(p)] = self._seen.get(id(p), 0) + 1
        if self._pending[bi] == 0:
            self._launch(bi)

    def _launch(self, bi):
        self._launched.add(bi)
        grads = [p.grad for p in self.buckets[bi] if p.grad is not None and id(p) not in self._early]
        if not grads:
            return
        flat = torch.cat([g.reshape(-1) for g in grads])
        h = dist.all_reduce(flat, async_op=True)
        self._handles.append((h, flat, grads))

    def finish(self):
        """Waits for the outstanding all-reduces, writes the averaged gradients back, re-arms the hooks."""
        W = world_size()
        # buckets holding parameters the step did not touch (unused heads: every rank runs the same task, so the same set)
        # never complete through the hooks - reduce whatever gradients they do hold now
        if is_dist():
            for bi in range(len(self.buckets)):
                if bi not in self._launched:
                    self._launch(bi)
        for h, flat, _keep in self._early_handles:
            h.wait()
            if flat is not None:     # gloo has no AVG: sum, then scale
                flat.div_(W)
        # the reduced slices ARE the parameters' gradients only if autograd adopted the arena views; where it made its own tensor
        # instead, hand the averaged values over (views of a slice start at 16-byte aligned offsets, see functional.GradArena)
        for flat, params in self._early_slices:
            o = 0
            for p in params:
                n = p.numel()
                view = flat[o:o + n].view(p.shape)
                if not p.requires_grad:     # autograd left it without a gradient on purpose: nothing to hand over (an all-zero .grad would
                    pass                    # block the early path next step and feed weight decay / moments of an optimizer that owns it)
                elif p.grad is None:
                    p.grad = view
                elif p.grad.data_ptr() != view.data_ptr():
                    p.grad.copy_(view)
                o += (n + 3) // 4 * 4
        for h, flat, grads in self._handles:
            h.wait()
            flat.div_(W)
            o = 0
            for g in grads:
                n = g.numel()
                g.copy_(flat[o:o + n].view_as(g))
                o += n
        self.reset()
