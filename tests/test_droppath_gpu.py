"""Stochastic depth (train-mode DropPath, eva_vit_model.py:121-138) with INJECTED per-frame masks: the engine skips dropped
(block, branch, frame) triples entirely (functional.DropPlan: compacting LayerNorm gather, in-place scatter epilogue) while
the oracle evaluates every branch and multiplies by 0 or 1/keep as the reference does - values and gradients must agree.
Masks cover: every frame kept in a branch (out-of-place path), a branch with no frame kept, mixed."""
import pytest
import torch

from common import build_model, rel_err
from mico_amd import runtime
from mico_amd.weights import synth_inputs
from oracle import mico_oracle as O

pytestmark = pytest.mark.gpu


def _masks(depth, B, keep, seed):
    g = torch.Generator().manual_seed(seed)
    m = (torch.rand(depth, 2, B, generator=g) < keep).float() / keep
    m[0, 0] = 1.0 / keep          # block 0 attention: all kept  -> out-of-place path with a scale
    if depth > 1:
        m[1, 1] = 0.0             # block 1 MLP: nothing kept   -> branch skipped
        m[1, 0, 0] = 0.0
        m[1, 0, 1:] = 1.0 / keep
    return m


@pytest.mark.parametrize("vtype", ["evaclip02_base", "evaclip01_giant"])
def test_tower_frame_skipping(cuda, vtype):
    torch.set_num_threads(16)
    depth, B = 3, 5
    m, sd = build_model(vtype, depth, device=cuda)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, 3, 224, 224, generator=g)
    dps = _masks(depth, B, 0.7, 11)
    pre = "vision_encoder.visual."
    names = [pre + "blocks.0.norm1.weight", pre + "blocks.1.mlp.%s.weight" % ("w3" if "02" in vtype else "fc2"), pre + "pos_embed",
             pre + "blocks.2.attn.%s" % ("q_proj.weight" if "02" in vtype else "qkv.weight"), pre + "patch_embed.proj.weight",
             pre + "blocks.1.norm2.weight", pre + "blocks.2.attn.proj.bias"]
    sdo = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
    ref = O.eva_vit_forward(sdo, x, O.ARCHS[vtype], drop_path_scale=dps)
    w = torch.randn(ref.shape, generator=g) / ref.numel() ** 0.5
    (ref * w).sum().backward()
    vis = m.vision_encoder.visual
    with runtime.precision(torch.float16):
        m.zero_grad(set_to_none=True)
        out = vis.forward_groups([x.to(cuda)], drop_path_scale=dps)
        (out * w.to(cuda)).sum().backward()
    e = rel_err(out, ref)
    print(vtype, "tokens", f"{e:.2e}")
    assert e < 1e-3
    named = dict(m.named_parameters())
    for n in names:
        if named[n].grad is None:      # parameters of a branch no frame went through get no gradient at all
            assert sdo[n].grad.abs().max() == 0, n
            continue
        ge = rel_err(named[n].grad, sdo[n].grad)
        print(" ", n, f"{ge:.2e}")
        assert ge < 2e-2, (n, ge)
    # the skipped MLP branch of block 1 must leave its parameters without a gradient contribution
    assert sdo[pre + "blocks.1.norm2.weight"].grad.abs().max() == 0
    gskip = named[pre + "blocks.1.norm2.weight"].grad
    assert gskip is None or gskip.abs().max() == 0


def test_alignment_step_with_masks(cuda):
    """Whole alignment step (ITC+ITM+CAP) in TRAIN mode - stochastic depth with per-modality injected masks and BERT dropout with
    injected per-pass seeds - in the fp16 parity configuration."""
    torch.set_num_threads(16)
    vtype, depth, b = "evaclip01_giant", 2, 3
    m, sd = build_model(vtype, depth, device=cuda)
    inp = synth_inputs(dict(b=b, vision=2, audio=1, S=12), seed=77)
    import random
    mi, lab = O.token_masker(inp["input_ids"], 0.6, random.Random(0))
    idx = torch.arange(b).roll(1)
    dps = {"v": _masks(depth, b * 2, 0.6, 3), "a": _masks(depth, b, 0.6, 4)}
    inj = {"tva": dict(neg_cond_idx=idx, neg_text_idx=idx), "cap": dict(masked_ids=mi, labels=lab), "drop_path_scale": dps}
    sdo = dict(sd)
    sdo["multimodal_encoder.cls.predictions.decoder.weight"] = sdo["multimodal_encoder.bert.embeddings.word_embeddings.weight"]
    arch = O.ARCHS[vtype]
    seeds = [11, 22, 33, 44, 55]          # one per BERT pass, consumed in call order: text, ITM, CAP
    with torch.no_grad(), O.bert_dropout(0.1, 0.1, seeds):
        ref, _ = O.mico_forward(sdo, arch, inp, "ret%tva_cap%tva", dict(itm_ratio=0.1), injected=inj)
    batch = {k: v.to(cuda) for k, v in inp.items()}
    batch["_injected"] = inj
    m.train()
    m.multimodal_encoder.bert.dropout_seed_source = iter(seeds).__next__
    try:
        with runtime.precision(torch.float16):
            out = m(batch, "ret%tva_cap%tva", compute_loss=True)
    finally:
        m.multimodal_encoder.bert.dropout_seed_source = None
    for k, v in ref.items():
        e = abs(out[k].item() - v.item()) / max(abs(v.item()), 1e-6)
        print(k, out[k].item(), v.item(), f"{e:.2e}")
        assert e < 2e-3, k


def test_chunked_tower_recompute(cuda):
    """Tower passes in chunks (forward without saving, per-chunk recompute + backward; configs[3]-sized inputs need this):
    same values and gradients as the single-pass schedule, with and without stochastic depth, across mixed modality groups."""
    from mico_amd import runtime as rt
    depth = 2
    m, sd = build_model("evaclip02_base", depth, device=cuda)
    vis = m.vision_encoder.visual
    g = torch.Generator().manual_seed(9)
    img = torch.randn(4, 3, 224, 224, generator=g).to(cuda)
    aud = torch.randn(3, 1, 224, 224, generator=g).to(cuda)
    dps = _masks(depth, 7, 0.7, 21)
    w = (torch.randn(7, 197, 768, generator=g) / (7 * 197 * 768) ** 0.5).to(cuda)
    res = {}
    for chunk in (None, 3, 2):
        rt.set_tower_chunk(chunk)
        try:
            for use_dp in (False, True):
                with rt.precision(torch.float16):
                    m.zero_grad(set_to_none=True)
                    out = vis.forward_groups([img, aud], drop_path_scale=dps if use_dp else None)
                    (out * w).sum().backward()
                grads = {n: p.grad.clone() for n, p in vis.named_parameters() if p.grad is not None}
                res[(chunk, use_dp)] = (out.detach().clone(), grads)
        finally:
            rt.set_tower_chunk(None)
    for use_dp in (False, True):
        o0, g0 = res[(None, use_dp)]
        for chunk in (3, 2):
            o1, g1 = res[(chunk, use_dp)]
            assert torch.equal(o0, o1), (chunk, use_dp)          # per-frame arithmetic is batch independent
            assert set(g0) == set(g1)
            for n in g0:
                assert rel_err(g1[n], g0[n]) < 2e-3, (chunk, use_dp, n, rel_err(g1[n], g0[n]))


def test_activation_diet_recompute(cuda):
    """runtime.set_activation_diet: level 1 drops the MLP intermediates (fc1 + GELU / GELU' re-run in the backward), level 2 the LayerNorm
    outputs too (recomputed from the saved fp32 rows).  Plain-MLP tower (g/14 architecture), with and without stochastic depth (compact
    kept-frame rows vs the whole stream as the saved LayerNorm input), also through chunks: the recomputed tensors are the forward's own
    values, so outputs are identical and gradients agree to the summation order of the fp32 atomics.  Level 3 (round 5): the LayerNorm input
    rows kept as fp16 normalised rows instead of fp32 copies, with the MLP intermediates of none / the last / all blocks kept (the mixed
    per-block plan tower_plan produces for one rank of configs[3]) - same outputs, gradients within the same bound."""
    from mico_amd import runtime as rt
    from mico_amd import functional as Fn
    depth = 2
    m, sd = build_model("evaclip01_giant", depth, device=cuda)
    vis = m.vision_encoder.visual
    g = torch.Generator().manual_seed(11)
    img = torch.randn(3, 3, 224, 224, generator=g).to(cuda)
    aud = torch.randn(2, 1, 224, 224, generator=g).to(cuda)
    dps = _masks(depth, 5, 0.7, 23)
    w = (torch.randn(5, 257, 1408, generator=g) / (5 * 257 * 1408) ** 0.5).to(cuda)
    res = {}
    try:
        cases = ((0, None, 0), (1, None, 0), (2, None, 0), (2, 2, 0), (3, None, 0), (3, None, 1), (3, None, 2), (3, 2, 1))
        for diet, chunk, mlpb in cases:
            rt.set_activation_diet(diet, mlpb)
            rt.set_tower_chunk(chunk)
            for use_dp in (False, True):
                with rt.precision(torch.float16):
                    m.zero_grad(set_to_none=True)
                    out = vis.forward_groups([img, aud], drop_path_scale=dps if use_dp else None)
                    assert rt.last_tower_plan["diet"] == diet and rt.last_tower_plan["rows_fp16_normalised"] == (diet == 3)
                    assert rt.last_tower_plan["mlp_blocks_kept"] == (depth if diet == 0 else mlpb)
                    (out * w).sum().backward()
                res[(diet, chunk, mlpb, use_dp)] = (out.detach().clone(), {n: p.grad.clone() for n, p in vis.named_parameters() if p.grad is not None})
    finally:
        rt.set_activation_diet(None)
        rt.set_tower_chunk(None)
    worst = 0.0
    for use_dp in (False, True):
        o0, g0 = res[(0, None, 0, use_dp)]
        for key in cases[1:]:
            o1, g1 = res[key + (use_dp,)]
            assert torch.equal(o0, o1), (key, use_dp)
            assert set(g0) == set(g1)
            for n in g0:
                assert rel_err(g1[n], g0[n]) < 2e-3, (key, use_dp, n, rel_err(g1[n], g0[n]))
                if key[0] == 3:
                    worst = max(worst, rel_err(g1[n], g0[n]))
    print("level 3 (fp16 normalised rows) against level 0: worst gradient difference", worst)
    # round 5 (late): at level 3 the qkv / kept-block fc1 weight gradients are taken against the normalised rows and norm1 / norm2's gamma, beta
    # applied to the result (functional._wgrad_ln_folded) - against the same level with the LayerNorm outputs re-created first
    # (runtime.CFG.ln_fold_wgrad off): the same sums up to the fp16 rounding of the re-created output that the folded form does not make
    assert rt.CFG.ln_fold_wgrad
    with torch.no_grad():       # (LayerNorms start as the identity map: give the fold something to fold)
        gp = torch.Generator().manual_seed(3)
        for n, p in vis.named_parameters():
            if ".norm1." in n or ".norm2." in n:
                p.add_((torch.randn(p.shape, generator=gp) * 0.3).to(cuda))
    both, calls, real_fold = {}, {True: 0, False: 0}, Fn.ops.dw_colfold
    try:
        rt.set_activation_diet(3, 1)
        for fold in (True, False):
            rt.CFG.ln_fold_wgrad = fold

            def counted(*a, _f=fold, **k):
                calls[_f] += 1
                return real_fold(*a, **k)
            Fn.ops.dw_colfold = counted
            with rt.precision(torch.float16):
                m.zero_grad(set_to_none=True)
                out = vis.forward_groups([img, aud], drop_path_scale=None)      # (this draw of `dps` drops the last block's MLP branch in every frame)
                (out * w).sum().backward()
            both[fold] = (out.detach().clone(), {n: p.grad.clone() for n, p in vis.named_parameters() if p.grad is not None})
    finally:
        Fn.ops.dw_colfold = real_fold
        rt.CFG.ln_fold_wgrad = True
        rt.set_activation_diet(None)
    (o1, g1), (o2, g_plain) = both[True], both[False]
    assert torch.equal(o1, o2)
    fold_worst = max(rel_err(g1[n], g_plain[n]) for n in g_plain)
    differs = [n for n in g_plain if not torch.equal(g1[n], g_plain[n])]
    print("folded against re-created LayerNorm outputs: worst", fold_worst, "in", len(differs), "tensors;", calls[True], "folded weight gradients")
    assert fold_worst < 1e-3 and any("qkv.weight" in n for n in differs) and any("fc1.weight" in n for n in differs)
    # qkv of both blocks + fc1 of the one block that kept its MLP intermediates (per tower pass), and none with the switch off
    assert calls[True] >= depth + 1 and calls[True] % (depth + 1) == 0 and calls[False] == 0, calls
    # the plan: the cheapest combination that fits - everything fits here, so nothing is dropped
    spec, _ = vis._tower_spec()
    chunk0, diet0 = Fn.tower_plan(spec, 5, cuda)
    assert (chunk0, diet0.level, diet0.xh16, all(diet0.keep_mlp), all(diet0.keep_ln)) == (5, 0, False, True, True)
    # and a step that cannot keep everything: one rank of BASELINE configs[3] on the full-depth architecture (896 frames, ~0.8 kept) - priced
    # from the free memory of this box, it must come out as a single pass on the diet or as chunks, never as "keep everything in one pass"
    import copy
    big = copy.copy(spec)
    big.arch = dict(spec.arch, depth_built=40)
    chunk, diet = Fn.tower_plan(big, 896, cuda, kept=0.8)
    assert diet.level > 0 or chunk < 896          # 896 x 0.82 x 542 MB = 398 GB of level-0 activations exceed the 288 GB of an MI355X
    per_tok = {0: 20 * 1408 + 4 * 6144, 1: 20 * 1408, 2: 16 * 1408, 3: 12 * 1408}[diet.level]
    mlp_b = 2 if diet.mlp_pre else 4       # a kept block keeps its fc1 pre-activation (round 6: where memory limits the kept blocks) or gelu + gelu'
    assert diet.mlp_pre == (Fn._mlp_keeps_pre() and diet.level == 3 and diet.mlp_blocks > 0)      # 896 frames never fit 40 pairs
    acts = chunk * 0.82 * 257 * (40 * per_tok + (diet.mlp_blocks * mlp_b * 6144 if diet.level == 3 else 0))
    assert acts < torch.cuda.mem_get_info(cuda)[1]
    print("configs[3] rank share on this box:", chunk, "frames per pass, diet", diet.describe())
    # the soft budget (functional._SOFT_FRAC of the device memory for the step's projected peak): on a clean 288 GiB device level 1 would
    # "fit" at a 266 GiB peak - the plan must stay in ONE pass under the soft budget: level 3 (fp16 normalised rows) with the MLP intermediates
    # of SOME blocks kept from what the smaller rows free (round 4: level 2, 231 GiB measured), leaving room for RCCL and a second reducer
    if torch.cuda.mem_get_info(cuda)[0] > 250 << 30:
        assert (chunk, diet.level) == (896, 3) and 4 <= diet.mlp_blocks <= (40 if mlp_b == 2 else 24), (chunk, diet.describe())
        if mlp_b == 2 and diet.mlp_blocks < 40:      # what is dropped is never a head-split block of the timed precision (their fc1 recompute costs twice)
            from common import precision_config
            with precision_config("timed"):
                _, dh = Fn.tower_plan(big, 896, cuda, kept=0.8)
            assert all(dh.keep_mlp[:4]) and dh.mlp_blocks >= diet.mlp_blocks - 1, dh.describe()
        proj = torch.cuda.memory_allocated(cuda) + (12 << 30) + 896 * (52 << 20) + acts
        assert proj < Fn._SOFT_FRAC * torch.cuda.mem_get_info(cuda)[1]
        # with the per-block kept tokens of a real draw (later blocks keep fewer frames) the last blocks are cheaper than the average
        toks = [int(896 * 257 * (1.0 - 0.4 * i / 39)) for i in range(40)]
        chunk_b, diet_b = Fn.tower_plan(big, 896, cuda, kept=sum(toks) / (40 * 896 * 257), block_tokens=toks)
        assert (chunk_b, diet_b.level) == (896, 3) and diet_b.mlp_blocks >= diet.mlp_blocks


def test_tower_forward_oom_retry(cuda, monkeypatch):
    """An out-of-memory error in the single-pass tower forward (the plan's fitted headroom under-estimated the step) is answered ONCE by the
    next more conservative plan - the next activation-diet level, or two chunks when the diet is exhausted - instead of failing the step:
    same output, same gradients, the decision recorded in runtime.last_tower_plan."""
    from mico_amd import runtime as rt
    from mico_amd import functional as Fn
    depth = 2
    m, sd = build_model("evaclip01_giant", depth, device=cuda)
    vis = m.vision_encoder.visual
    g = torch.Generator().manual_seed(12)
    img = torch.randn(4, 3, 224, 224, generator=g).to(cuda)
    w = (torch.randn(4, 257, 1408, generator=g) / (4 * 257 * 1408) ** 0.5).to(cuda)
    dps = _masks(depth, 4, 0.7, 29)

    def run():
        with rt.precision(torch.float16):
            m.zero_grad(set_to_none=True)
            out = vis.forward_groups([img], drop_path_scale=dps)
            (out * w).sum().backward()
        return out.detach().clone(), {n: p.grad.clone() for n, p in vis.named_parameters() if p.grad is not None}

    o0, g0 = run()
    assert rt.last_tower_plan["diet"] == 0 and "oom_retry" not in rt.last_tower_plan
    real = Fn._tower_forward
    for forced_diet, want in ((None, dict(diet=1, frames_per_pass=4)), (2, dict(diet=2, frames_per_pass=2)), (3, dict(diet=3, frames_per_pass=2))):
        fails = [1]
        mem = {}

        def flaky(*a, **k):
            if fails[0]:
                fails[0] -= 1
                mem["before"] = torch.cuda.memory_allocated()
                attempt = real(*a, **k)      # the error comes LATE: every buffer of the failed attempt is alive in this frame when it leaves it
                mem["at_raise"] = torch.cuda.memory_allocated()
                raise torch.cuda.OutOfMemoryError("injected")
            mem.setdefault("at_retry", torch.cuda.memory_allocated())      # (the first call after the failure: chunks and the backward's recompute follow)
            return real(*a, **k)

        monkeypatch.setattr(Fn, "_tower_forward", flaky)
        rt.set_activation_diet(forced_diet)
        try:
            o1, g1 = run()
        finally:
            rt.set_activation_diet(None)
            monkeypatch.setattr(Fn, "_tower_forward", real)
        plan = rt.last_tower_plan
        assert plan.get("oom_retry") and all(plan[k] == v for k, v in want.items()), plan
        # the retry must not run on top of the failed attempt's activations (the handler's traceback kept them alive)
        assert mem["at_raise"] - mem["before"] > 20 << 20, mem                    # (the failed attempt held its saved activations)
        assert mem["at_retry"] - mem["before"] < 0.25 * (mem["at_raise"] - mem["before"]), mem
        assert rel_err(o1, o0) < 1e-6       # (chunks draw per-chunk plans from the same masks: the same frames are kept)
        for n in g0:
            assert rel_err(g1[n], g0[n]) < 2e-3, (n, rel_err(g1[n], g0[n]))
