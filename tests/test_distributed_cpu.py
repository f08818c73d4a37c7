"""world_size-2 gloo tests (CPU) of the data-parallel exchange steps in mico_amd/distributed.py: the packed all-gather, the
index-then-fetch row exchange with its mirrored gradient route (checked against the reference's all_gather_with_grad
semantics), the bucketed gradient reducer, and the ITC target / own-diagonal bookkeeping per rank."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fn(rank, world)
        ret[rank] = "ok"
    except Exception as e:   # noqa
        import traceback
        ret[rank] = traceback.format_exc()
    finally:
        dist.destroy_process_group()


def run2(fn, world=2):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), fn, ret), nprocs=world, join=True)
    for r in range(world):
        assert ret.get(r) == "ok", ret.get(r)


def _packed(rank, world):
    from mico_amd import distributed as D
    g = torch.Generator().manual_seed(rank)
    b = 3
    feat = torch.randn(b, 8, generator=g)
    ids = torch.randint(0, 1000, (b, 5), generator=g)
    mask = torch.randint(0, 2, (b, 5), generator=g)
    f_all, i_all, m_all = D.packed_all_gather([feat, ids, mask])
    assert f_all.shape == (world * b, 8) and i_all.dtype == torch.int64
    assert torch.equal(f_all, D.concat_all_gather(feat))
    assert torch.equal(i_all, D.concat_all_gather(ids)) and torch.equal(m_all, D.concat_all_gather(mask))
    assert torch.equal(f_all[rank * b:(rank + 1) * b], feat)
    assert not f_all.requires_grad


def _fetch(rank, world):
    from mico_amd import distributed as D
    g = torch.Generator().manual_seed(10 + rank)
    b, E, Dm = 4, 3, 5
    cond = torch.randn(b, E, Dm, generator=g, requires_grad=True)
    idx = torch.randint(0, world * b, (b,), generator=g)
    wts = torch.randn(b, E, Dm, generator=g)
    # reference semantics: all_gather_with_grad(cond)[idx]  (data/model/vast.py:421-433)
    ref_all = D.all_gather_with_grad(cond)
    ref = ref_all[idx]
    (ref * wts).sum().backward()
    gref = cond.grad.clone()
    cond.grad = None
    out = D.fetch_rows(cond, idx)
    assert torch.equal(out, ref.detach())
    (out * wts).sum().backward()
    assert torch.allclose(cond.grad, gref, atol=1e-6), (cond.grad - gref).abs().max()


def _reducer(rank, world):
    from mico_amd.distributed import GradBucketReducer
    torch.manual_seed(0)
    lin = torch.nn.Sequential(torch.nn.Linear(6, 7), torch.nn.Linear(7, 3))
    red = GradBucketReducer(lin.parameters(), bucket_bytes=64)   # several tiny buckets
    assert len(red.buckets) >= 2
    for step in range(2):
        g = torch.Generator().manual_seed(100 * step + rank)
        x = torch.randn(5, 6, generator=g)
        lin.zero_grad(set_to_none=True)
        lin(x).square().sum().backward()
        local = [p.grad.clone() for p in lin.parameters()]
        red.finish()
        for p, l in zip(lin.parameters(), local):
            gs = [torch.empty_like(l) for _ in range(world)]
            dist.all_gather(gs, l)
            assert torch.allclose(p.grad, sum(gs) / world, atol=1e-6)


def _staged_reducer(rank, world):
    """The staged step's contract with the reducer (round 6, MiCo.forward(backward_scale=...)): a "head" that is differentiated INSIDE the forward,
    twice (two condition sets), against a detached leaf of the "tower" output - its parameters accumulate gradients before the step's one real
    backward, one of them (shared with the main graph, like BERT's text pass) a third time in it - and the tower gets the leaf's gradient through
    mico_forward._StagedLoss.  The reducer's bucket hooks must ignore the staged accumulations (distributed.staged_backward) and the averaged
    gradients must equal those of the direct form, also with a loss scale other than 1."""
    from mico_amd.distributed import GradBucketReducer, staged_backward
    from mico_amd.model.mico_forward import _StagedLoss
    torch.manual_seed(0)
    tower = torch.nn.Linear(6, 5)
    shared = torch.nn.Linear(5, 4)        # used by the main graph AND by the staged heads
    head = torch.nn.Linear(4, 1)          # staged only
    params = list(tower.parameters()) + list(shared.parameters()) + list(head.parameters())
    g = torch.Generator().manual_seed(7 + rank)
    x = torch.randn(9, 6, generator=g)

    def losses(feat, sets):
        main = shared(feat).square().mean()
        staged = [head(torch.tanh(shared(feat * (k + 1)))).square().mean() for k in range(sets)]
        return main, staged

    def direct(scale):
        for p in params:
            p.grad = None
        main, staged = losses(tower(x), 2)
        ((main + sum(staged)) * scale).backward()

    red = GradBucketReducer(params, bucket_bytes=64)
    direct(1.0)
    red.finish()
    want = [p.grad.clone() for p in params]
    for scale in (1.0, 8.0):
        for p in params:
            p.grad = None
        feat = tower(x)
        leaf = feat.detach().requires_grad_(True)
        total = None
        for k in range(2):
            lk = head(torch.tanh(shared(leaf * (k + 1)))).square().mean()
            with staged_backward():
                torch.autograd.backward(lk * scale)
            total = lk.detach() if total is None else total + lk.detach()
        assert head.weight.grad is not None and tower.weight.grad is None
        out = _StagedLoss.apply(total, float(scale), 1, feat, leaf.grad)
        main = shared(feat).square().mean()
        ((main + out) * scale).backward()
        red.finish()
        for p, w in zip(params, want):
            assert torch.allclose(p.grad / scale, w, atol=1e-6, rtol=1e-5), (scale, (p.grad / scale - w).abs().max())
    red.close()


class _ArenaFn(torch.autograd.Function):
    """Stand-in for the ViT tower's backward: parameter gradients live in one flat arena, the finished slice is announced to
    runtime.grad_slice_hook() from INSIDE the backward (functional._tower_backward does this per block)."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return x @ w.t() + b

    @staticmethod
    def backward(ctx, dy):
        from mico_amd import runtime
        from mico_amd.functional import GradArena
        x, w = ctx.saved_tensors
        arena = GradArena([w, dy.new_empty(w.shape[0])])
        arena.get(0).copy_(dy.t() @ x)
        arena.get(1).copy_(dy.sum(0))
        hook = runtime.grad_slice_hook()
        if hook is not None:
            hook(arena.span(0, 2), [ctx.w_param, ctx.b_param])
        return dy @ w, arena.get(0), arena.get(1)


def _arena_reducer(rank, world):
    """The early in-place all-reduce of an arena slice: right when autograd adopts the arena views (no prior .grad), and
    left to the bucket path when a .grad already exists (set_to_none=False / accumulation); a second backward before finish() raises."""
    from mico_amd.distributed import GradBucketReducer
    torch.manual_seed(0)
    lin = torch.nn.Linear(6, 5)
    red = GradBucketReducer(lin.parameters(), bucket_bytes=1 << 20)

    def run(x):
        fn_ctx = {}
        class F(_ArenaFn):
            @staticmethod
            def backward(ctx, dy):
                ctx.w_param, ctx.b_param = lin.weight, lin.bias
                return _ArenaFn.backward(ctx, dy)
        F.apply(x, lin.weight, lin.bias).square().sum().backward()

    def expect(x_of_rank, prior=None):
        tot = [torch.zeros_like(p) for p in lin.parameters()]
        for r in range(world):
            xr = x_of_rank(r)
            y = (xr @ lin.weight.detach().t() + lin.bias.detach())
            dy = 2 * y
            tot[0] += dy.t() @ xr
            tot[1] += dy.sum(0)
        out = [t / world for t in tot]
        if prior is not None:
            out = [o + p for o, p in zip(out, prior)]
        return out

    xs = lambda step: (lambda r: torch.randn(4, 6, generator=torch.Generator().manual_seed(50 * step + r)))
    # (1) no prior gradient: early path, views adopted
    lin.zero_grad(set_to_none=True)
    run(xs(0)(rank))
    assert len(red._early_slices) == 1
    red.finish()
    for p, e in zip(lin.parameters(), expect(xs(0))):
        assert torch.allclose(p.grad, e, atol=1e-5), (p.grad - e).abs().max()
    # (2) an existing .grad (zeros from set_to_none=False): the early path must stand back, the bucket path reduces the sum
    lin.zero_grad(set_to_none=False)
    run(xs(1)(rank))
    assert len(red._early_slices) == 0
    red.finish()
    for p, e in zip(lin.parameters(), expect(xs(1))):
        assert torch.allclose(p.grad, e, atol=1e-5), (p.grad - e).abs().max()
    # (3) two backward passes before finish(): refused loudly instead of leaving the second contribution rank-local
    lin.zero_grad(set_to_none=True)
    run(xs(2)(rank))
    try:
        run(xs(3)(rank))
        raise AssertionError("expected the double-accumulation guard to fire")
    except RuntimeError as e:
        assert "accumulated twice" in str(e)
    red.reset()
    from mico_amd import runtime
    runtime.set_grad_slice_hook(None)


def _targets(rank, world):
    """ITC targets and the own-rank diagonal (vast.py:409-427) as MiCo.forward computes them, against a single-process
    evaluation on the concatenated global batch."""
    from mico_amd import distributed as D
    b, d = 3, 8
    g = torch.Generator().manual_seed(5)
    ft_all = torch.nn.functional.normalize(torch.randn(world * b, d, generator=g), dim=-1)
    fc_all = torch.nn.functional.normalize(torch.randn(world * b, d, generator=g), dim=-1)
    ft, fc = ft_all[rank * b:(rank + 1) * b], fc_all[rank * b:(rank + 1) * b]
    got_t, got_c = D.packed_all_gather([ft, fc])
    assert torch.equal(got_t, ft_all) and torch.equal(got_c, fc_all)
    targets = torch.arange(D.rank() * b, D.rank() * b + b)
    sim = ft @ got_c.t() / 0.07
    full = ft_all @ fc_all.t() / 0.07
    assert torch.allclose(sim, full[rank * b:(rank + 1) * b])
    loss = torch.nn.functional.cross_entropy(sim, targets, label_smoothing=0.1)
    ref = torch.nn.functional.cross_entropy(full, torch.arange(world * b), label_smoothing=0.1, reduction="none")
    assert torch.allclose(loss, ref[rank * b:(rank + 1) * b].mean(), atol=1e-6)
    w = torch.softmax(sim, 1) + 1e-4
    w[:, rank * b: rank * b + b].fill_diagonal_(0)
    assert (w[torch.arange(b), targets] == 0).all() and (w > 0).sum() == b * (world * b - 1)


@pytest.mark.parametrize("fn", [_packed, _fetch, _reducer, _staged_reducer, _arena_reducer, _targets])
def test_world2_gloo(fn):
    run2(fn)


def test_single_process_identities():
    from mico_amd import distributed as D
    x = torch.randn(4, 3, requires_grad=True)
    assert D.world_size() == 1 and D.rank() == 0
    assert torch.equal(D.concat_all_gather(x), x) and not D.concat_all_gather(x).requires_grad
    assert D.all_gather_with_grad(x) is x
    idx = torch.tensor([2, 0, 2, 1])
    out = D.fetch_rows(x, idx)
    out.sum().backward()
    assert torch.equal(x.grad[:, 0], torch.tensor([1.0, 1.0, 2.0, 0.0]))
