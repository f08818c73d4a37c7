"""Pins the CPU oracle (oracle/mico_oracle.py) against fixtures produced by the REFERENCE itself (oracle/make_golden.py,
build container).  fp32 vs fp32: the bar is 2e-5 relative.  Everything here runs on CPU."""
import pytest
import torch
import torch.nn.functional as F

from common import golden, rel_err, build_model, grad_digest_check
from oracle import mico_oracle as O
from mico_amd.weights import synth_inputs

TOL = 2e-5


@pytest.fixture(scope="module", params=[("evaclip02_base", "b16_d2"), ("evaclip01_giant", "g14_d2")])
def setup(request):
    vtype, tag = request.param
    m, sd = build_model(vtype, 2)
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    # weight tying of the LM decoder (transformers==4.31 behaviour of the reference, model/bert.py:1038-1041)
    sd["multimodal_encoder.cls.predictions.decoder.weight"] = sd["multimodal_encoder.bert.embeddings.word_embeddings.weight"]
    sd["multimodal_encoder.cls.predictions.decoder.bias"] = sd["multimodal_encoder.cls.predictions.bias"]
    return vtype, tag, sd, O.ARCHS[vtype]


def test_vit_tower(setup):
    vtype, tag, sd, arch = setup
    fx = golden(f"vit_{tag}.pt")
    g = torch.Generator().manual_seed(fx["meta"]["input_seed"])
    x = torch.randn((2, 3, 224, 224), generator=g)
    taps = []
    out = O.eva_vit_forward(sd, x, arch, taps=taps)
    assert rel_err(out, fx["out"]) < TOL
    assert rel_err(torch.stack([t.mean() for t in taps]), fx["tap_mean"]) < 1e-4
    assert rel_err(torch.stack([t[:, [0, 1, 100]] for t in taps]), fx["tap_rows"]) < TOL
    w = torch.randn(out.shape, generator=g) / out.numel() ** 0.5
    (out * w).sum().backward()
    for n, d in fx["grads"].items():
        assert grad_digest_check(d, sd["vision_encoder.visual." + n].grad, TOL) < 5e-5, n


def test_vit_tower_large():
    """EVA02-CLIP-L/14 (mico.py:336-340; depth 2): the tower variant with the 2730-wide SwiGLU hidden layer."""
    m, sd = build_model("evaclip02_large", 2)
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    fx = golden("vit_l14_d2.pt")
    g = torch.Generator().manual_seed(fx["meta"]["input_seed"])
    x = torch.randn((2, 3, 224, 224), generator=g)
    out = O.eva_vit_forward(sd, x, O.ARCHS["evaclip02_large"])
    assert rel_err(out, fx["out"]) < TOL
    w = torch.randn(out.shape, generator=g) / out.numel() ** 0.5
    (out * w).sum().backward()
    for n, d in fx["grads"].items():
        assert grad_digest_check(d, sd["vision_encoder.visual." + n].grad, TOL) < 5e-5, n


def test_vit_tower_bige_postnorm():
    """EVA02-CLIP-bigE-14-plus (mico.py:341-344; depth 2): the POST-norm block order (eva_vit_model.py:411-413) against the reference's own
    EVAVisionTransformer (tests/golden/vit_bige_d2.pt: output, per-block taps, all 32 parameter-gradient digests)."""
    m, sd = build_model("evaclip02_bige", 2)
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    fx = golden("vit_bige_d2.pt")
    g = torch.Generator().manual_seed(fx["meta"]["input_seed"])
    x = torch.randn((2, 3, 224, 224), generator=g)
    taps = []
    out = O.eva_vit_forward(sd, x, O.ARCHS["evaclip02_bige"], taps=taps)
    assert rel_err(out, fx["out"]) < TOL
    assert rel_err(torch.stack([t[:, [0, 1, 100]] for t in taps]), fx["tap_rows"]) < TOL
    w = torch.randn(out.shape, generator=g) / out.numel() ** 0.5
    (out * w).sum().backward()
    assert len(fx["grads"]) == 32
    for n, d in fx["grads"].items():
        assert grad_digest_check(d, sd["vision_encoder.visual." + n].grad, TOL) < 5e-5, n


def swin_state_dict():
    """the synthetic weights of the fixture-sized Swin tower under the reference's own (prefix-free) SwinTransformer keys"""
    from mico_amd.model.swin import SWIN_CONFIGS, SwinTransformer
    from mico_amd.weights import synth_state_dict
    c = SWIN_CONFIGS["swin_tiny_test"]
    m = SwinTransformer(embed_dim=c["embed_dim"], depths=c["depths"], num_heads=c["num_heads"], drop_path_rate=0.0)
    return m, synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()})


def test_swin_tower():
    """model/swin.py SwinTransformer (row f4b) at fixture size: the oracle's restatement against the reference's own output, per-stage
    taps and all 119 parameter-gradient digests (tests/golden/swin_tiny.pt)."""
    m, sd = swin_state_dict()
    # the product module's buffers are its own arithmetic: they must equal what the reference computes with its loops
    for name, buf in m.named_buffers():
        if name.endswith("relative_position_index"):
            assert torch.equal(buf, O.swin_rel_index(7))
        elif name.endswith("attn_mask") and buf is not None:
            res = {0: 56, 1: 28, 2: 14}[int(name.split(".")[1])]
            assert torch.equal(buf, O.swin_shift_mask(res, 7, 3)), name
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    fx = golden("swin_tiny.pt")
    g = torch.Generator().manual_seed(fx["meta"]["input_seed"])
    x = torch.randn((2, 3, 224, 224), generator=g)
    taps = []
    out = O.swin_forward(sd, x, O.SWIN_ARCHS["swin_tiny_test"], pre="", taps=taps)
    assert out.shape == fx["out"].shape and rel_err(out, fx["out"]) < TOL
    # the reference's per-stage hook fires after the stage's PatchMerging; the oracle taps before it: compare the last stage (no merge)
    assert rel_err(taps[-1][:, [0, 1, 17]], fx["tap_rows"][-1]) < TOL
    w = torch.randn(out.shape, generator=g) / out.numel() ** 0.5
    (out * w).sum().backward()
    assert len(fx["grads"]) == 119
    for n, d in fx["grads"].items():
        assert grad_digest_check(d, sd[n].grad, TOL) < 5e-5, n


def test_bert(setup):
    vtype, tag, sd, arch = setup
    if tag != "b16_d2":
        pytest.skip("BERT fixture is tower independent")
    fx = golden("bert.pt")
    g = torch.Generator().manual_seed(fx["meta"]["seed"])
    b, S, E = fx["meta"]["b"], fx["meta"]["S"], fx["meta"]["E"]
    ids = torch.randint(1000, 30000, (b, S), generator=g)
    ids[:, 0] = 101
    mask = (torch.arange(S)[None] < fx["meta"]["lens"][:, None]).long()
    ids = ids * mask
    cond = torch.randn((b, E, 768), generator=g)
    for p in sd.values():
        p.grad = None
    o = O.bert_mlm(sd, ids, mask)
    assert rel_err(o["sequence_output"], fx["self_seq"]) < TOL
    assert torch.equal(o["logits"].argmax(-1), fx["self_argmax"])
    o = O.bert_mlm(sd, ids, mask, cond)
    assert rel_err(o["sequence_output"], fx["cross_seq"]) < TOL
    assert torch.equal(o["logits"].argmax(-1), fx["cross_argmax"])
    m3 = torch.tril(mask.unsqueeze(1).expand(-1, S, -1).clone())
    cr = cond.clone().requires_grad_(True)
    o = O.bert_mlm(sd, ids, m3, cr, fx["labels"])
    assert rel_err(o["sequence_output"], fx["causal_seq"]) < TOL
    assert abs(o["loss"].item() - fx["causal_loss"].item()) < 1e-5 * fx["causal_loss"].item()
    assert torch.equal(o["logits"].argmax(-1), fx["causal_argmax"])
    o["loss"].backward()
    assert rel_err(cr.grad, fx["causal_dcond"]) < 5e-5
    for n, d in fx["causal_grads"].items():
        key = "multimodal_encoder." + n
        gr = sd[key].grad
        if n == "bert.embeddings.word_embeddings.weight":   # tied with the LM decoder: same tensor object in sd
            pass
        assert grad_digest_check(d, gr, TOL) < 1e-4, n


def test_facade(setup):
    vtype, tag, sd, arch = setup
    fx = golden(f"facade_{tag}.pt")
    cfgs = {"n1": dict(b=2, vision=1, audio=1, depth=1, S=20), "n4": dict(b=2, vision=4, audio=4, depth=1, S=20),
            "n3": dict(b=2, vision=3, audio=2, depth=1, S=20)}
    with torch.no_grad():
        for name, c in cfgs.items():
            r = fx[name]
            inp = synth_inputs(c, seed=100)
            enc = O.encode_batch(sd, arch, inp)
            assert rel_err(enc["output_v"][:, :, [0, 1, 50]], r["vision_out_rows"]) < TOL
            assert rel_err(enc["output_a"][:, :, [0, 1, 50]], r["audio_out_rows"]) < TOL
            assert rel_err(enc["feat_t"], r["feat_t"]) < TOL
            for c_ in ("v", "a", "d", "va", "vd"):
                assert rel_err(O.feat_cond(sd, enc, c_), r["feat_" + c_]) < TOL, c_
            assert rel_err(enc["feat_t"] @ O.feat_cond(sd, enc, "v").t(), r["sim_t2v"]) < 1e-4
            for pv, k in ((False, "full"), (True, "pv")):
                cv = O.multimodal_input(sd, "vision", enc["output_v"], pv)
                ca = O.multimodal_input(sd, "audio", enc["output_a"], pv)
                cd = O.multimodal_input(sd, "depth", enc["output_d"], pv)
                assert rel_err(cv[:, [0, 1, cv.shape[1] - 1]], r[f"cond_v_{k}_rows"]) < TOL
                assert rel_err(cv.sum((1, 2)), r[f"cond_v_{k}_sum"]) < 1e-4
                assert rel_err(ca[:, [0, 1, ca.shape[1] - 1]], r[f"cond_a_{k}_rows"]) < TOL
                assert rel_err(cd[:, [0, 1, cd.shape[1] - 1]], r[f"cond_d_{k}_rows"]) < TOL
                out = O.bert_forward(sd, inp["input_ids"], inp["attention_mask"], cv)
                score = F.softmax(O.itm_head(sd, out[:, 0]), dim=1)[:, 1]
                assert rel_err(score, r[f"itm_score_{k}"]) < 1e-4


@pytest.mark.parametrize("W", [1, 2, 4])
def test_alignment_loss(setup, W):
    vtype, tag, sd, arch = setup
    fx = golden(f"loss_{tag}.pt")
    r = fx[f"W{W}"]
    b = fx["meta"]["b"]
    for p in sd.values():
        p.grad = None
    inputs = [synth_inputs(dict(b=b, vision=2, audio=1, S=12), seed=1234 + k) for k in range(W)]
    world = None
    enc_remote = None
    if W >= 2:
        with torch.no_grad():
            encs = [O.encode_batch(sd, arch, i) for i in inputs[1:]]
        enc_remote = {"condition_feats_" + c: torch.cat([O.condition_feats(e, c) for e in encs]) for c in ("v", "va")}
        world = dict(feat_t_all=r["world"]["feat_t_all"], ids_all=r["world"]["ids_all"], mask_all=r["world"]["mask_all"])
        for c in ("v", "va"):
            world[f"feat_{c}_all"] = r["world"][f"feat_{c}_all"]
        assert rel_err(enc_remote["condition_feats_va"].sum((1, 2)), r["remote_cond_va_sum"]) < 1e-4
    injected = {st: {k: r["inj"][st][k] for k in ("neg_cond_idx", "neg_text_idx")} for st in ("tva", "tv")}
    injected["cap"] = r["inj"]["cap"]
    cfg = dict(itm_ratio=fx["meta"]["itm_ratio"])
    if W >= 2:   # gathered condition memory = [local (with grad) | remote ranks (constant)]
        out, enc = _loss_w2(sd, arch, inputs[0], cfg, world, injected, enc_remote)
    else:
        out, enc = O.mico_forward(sd, arch, inputs[0], fx["meta"]["task"], cfg, injected=injected)
    for k, v in r["losses"].items():
        assert abs(out[k].item() - v.item()) < 2e-5 * max(1.0, abs(v.item())), k
    assert rel_err(enc["feat_t"], r["feat_t"]) < TOL
    sum(out.values()).backward()
    for n, d in r["grads"].items():
        assert grad_digest_check(d, sd[n].grad, TOL) < 2e-4, n


def _loss_w2(sd, arch, inp, cfg, world, injected, enc_remote):
    enc = O.encode_batch(sd, arch, inp)
    for c in ("v", "va"):
        world[f"cond_{c}_all"] = torch.cat((O.condition_feats(enc, c), enc_remote["condition_feats_" + c].detach()))
    ids, am = inp["input_ids"], inp["attention_mask"]
    l_itc, l_itm = [], []
    for st in ("tva", "tv"):
        c = st[1:]
        fc = O.feat_cond(sd, enc, c)
        li, _, _ = O.itc_loss(enc["feat_t"], fc, world["feat_t_all"], world[f"feat_{c}_all"], sd["contra_temp"], 0)
        l_itc.append(li)
        lm, _ = O.itm_loss(sd, ids, am, O.condition_feats(enc, c), world[f"cond_{c}_all"], world["ids_all"], world["mask_all"],
                           injected[st]["neg_cond_idx"], injected[st]["neg_text_idx"], cfg["itm_ratio"])
        l_itm.append(lm)
    cap = O.cap_loss(sd, injected["cap"]["masked_ids"], am, injected["cap"]["labels"], O.condition_feats(enc, "va"))
    return dict(loss_itc=sum(l_itc) / 2, loss_itm=sum(l_itm) / 2, loss_cap=cap), enc


def test_token_masker_matches_reference_rule():
    import random
    ids = torch.tensor([[101, 2000, 2001, 2002, 102, 0, 0], [101, 5, 102, 0, 0, 0, 0]])
    toks, labels = O.token_masker(ids, 0.6, random.Random(3))
    assert (labels[:, 0] == -100).all() and (labels[ids == 0] == -100).all()
    assert ((labels != -100).sum(1) >= 1).all()
    changed = toks != ids
    assert (labels[changed] == ids[changed]).all()


def test_token_masker_draw_for_draw_against_reference_class():
    """tests/golden/token_masker.pt: the reference's own TokenMasker (data/model/general_module.py:52-97) run under random.seed(s) by
    oracle/make_golden.py (`masker`).  O.token_masker with random.Random(s) - the same Mersenne-Twister stream - must produce the same
    masked ids and labels AND leave the generator at the same position (same number of draws in the same order: the retry loop of
    :71, one draw per non-pad position from column 1 on, then one kind draw per selected token and random.choice for the 10 % branch)."""
    import random
    fx = golden("token_masker.pt")
    assert len(fx["cases"]) >= 6
    for c in fx["cases"]:
        rng = random.Random(c["seed"])
        toks, labels = O.token_masker(c["ids"], c["p"], rng, mask_token=fx["meta"]["mask_token"],
                                      range_start=fx["meta"]["range_start"], range_end=fx["meta"]["range_end"])
        assert torch.equal(toks, c["masked"]) and torch.equal(labels, c["labels"]), c["seed"]
        assert rng.random() == c["next_draw"], c["seed"]
    # the uniform-number form the device kernel implements (mico_token_mask) is the same rule with the draws supplied as tensors:
    # feed it the reference's own draws, in the reference's order, and it must give the reference's result
    for c in fx["cases"][:4]:
        ids = c["ids"]
        b, S = ids.shape
        rng = random.Random(c["seed"])
        # replay general_module.py:69-74: row by row, rounds until something is selected
        per_row = []
        for i in range(b):
            rr = []
            while True:
                u = torch.ones(S)
                hit = False
                for j in range(1, S):
                    if ids[i, j] != 0:
                        u[j] = rng.random()
                        hit = hit or u[j] < c["p"]
                rr.append(u)
                if hit:
                    break
            per_row.append(rr)
        R = max(len(r) for r in per_row)
        u_mask = torch.ones(R, b, S)
        for i, rr in enumerate(per_row):
            for r, u in enumerate(rr):
                u_mask[r, i] = u
        u_kind, u_tok = torch.ones(b, S), torch.zeros(b, S)
        lo, hi = fx["meta"]["range_start"], fx["meta"]["range_end"]
        sel = c["labels"] != -100
        for i in range(b):
            for j in range(S):
                if sel[i, j]:
                    k = rng.random()
                    u_kind[i, j] = k
                    if 0.8 <= k < 0.9:
                        tok = rng.choice(list(range(lo, hi)))
                        u_tok[i, j] = (tok - lo + 0.5) / (hi - lo)
        toks, labels = O.token_masker_uniform(ids, c["p"], u_mask, u_kind, u_tok, mask_token=103, range_start=lo, range_end=hi)
        assert torch.equal(labels, c["labels"]), c["seed"]
        assert torch.equal(toks, c["masked"]), c["seed"]
