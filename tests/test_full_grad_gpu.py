"""Full-tensor gradient parity ("what's weak" 2): the golden fixtures keep a DIGEST of each reference gradient
(first 256 elements + L2 norm), so a wrong gradient confined to rows > 256 with a small norm share could pass them.  Here the CPU oracle -
pinned to the reference by those very digests, re-checked below - produces the complete gradient of EVERY parameter in-process and the HIP
product's gradient is compared element-wise: max|g - g_ref| / max|g_ref| per parameter, worst parameter printed.

Depth-2 towers (EVA02-B/16: RoPE + sub-LN + SwiGLU; EVA01-g/14: plain), BERT's causal cross-attention pass with the LM head, and the whole
alignment step (ITC + ITM + CAP)."""
import pytest
import torch

from common import golden, build_model, grad_digest_check
from mico_amd import runtime
from mico_amd.weights import synth_inputs
from oracle import mico_oracle as O

pytestmark = pytest.mark.gpu

TOL = 2e-2   # fp16 operands, fp32 accumulation; gradients are carried with a 4096x internal scale (tests/test_model_gpu.py)


def full_err(g, ref):
    g, ref = g.detach().float().cpu(), ref.detach().float()
    return ((g - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


def compare_all(named, ref_grads, tol, what):
    """named: product parameters by reference key; ref_grads: {key: full oracle gradient}.  Every key must be present on both sides."""
    errs = {}
    for n, gr in ref_grads.items():
        assert n in named and named[n].grad is not None, f"{what}: the product has no gradient for {n}"
        assert tuple(named[n].grad.shape) == tuple(gr.shape), (n, named[n].grad.shape, gr.shape)
        if n.endswith("self.key.bias"):
            # softmax is invariant to a shift of all of a row's scores, so the key bias has NO gradient: the reference's is pure fp32
            # rounding noise (~1e-9) and the product's fp16-operand noise; both must vanish against the sibling query bias gradient
            qs = ref_grads[n.replace("self.key.bias", "self.query.bias")].abs().max()
            # (the product's value is rounding noise whose size follows the accumulation order: 2.02e-3 of the query-bias scale on one box since the
            # BERT passes of a step share one gradient arena, below 2e-3 before and on other boxes)
            assert gr.abs().max() < 1e-5 * qs and named[n].grad.abs().max().item() < 4e-3 * qs.item(), (n, gr.abs().max(), named[n].grad.abs().max(), qs)
            continue
        errs[n] = full_err(named[n].grad, gr)
    worst = max(errs, key=errs.get)
    print(f"{what}: {len(errs)} parameters compared element-wise, worst {worst} = {errs[worst]:.2e}, median {sorted(errs.values())[len(errs) // 2]:.2e}")
    bad = {n: e for n, e in errs.items() if not e < tol}
    assert not bad, bad
    return errs


@pytest.fixture(scope="module", params=[("evaclip02_base", "b16_d2"), ("evaclip01_giant", "g14_d2")])
def setup(request, cuda):
    vtype, tag = request.param
    m, sd = build_model(vtype, 2, device=cuda)
    return vtype, tag, m, sd


def oracle_sd(sd):
    sdo = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    sdo["multimodal_encoder.cls.predictions.decoder.weight"] = sdo["multimodal_encoder.bert.embeddings.word_embeddings.weight"]
    return sdo


def test_vit_tower_full_gradients(setup, cuda):
    vtype, tag, m, sd = setup
    fx = golden(f"vit_{tag}.pt")
    g = torch.Generator().manual_seed(fx["meta"]["input_seed"])
    x = torch.randn((2, 3, 224, 224), generator=g)
    w = torch.randn(fx["out"].shape, generator=g) / fx["out"].numel() ** 0.5
    # oracle: complete gradients, and the oracle itself against the reference's digests (the pin)
    sdo = oracle_sd(sd)
    ref = O.eva_vit_forward(sdo, x, O.ARCHS[vtype])
    (ref * w).sum().backward()
    pre = "vision_encoder.visual."
    for n, d in fx["grads"].items():
        assert grad_digest_check(d, sdo[pre + n].grad, None) < 1e-4, ("oracle vs reference digest", n)
    ref_grads = {k[len(pre):]: v.grad for k, v in sdo.items() if k.startswith(pre) and v.requires_grad and v.grad is not None}
    assert len(ref_grads) >= 20
    m.zero_grad(set_to_none=True)
    with runtime.precision(torch.float16):
        out = m.vision_encoder.visual(x.to(cuda), return_all_features=True)
        (out * w.to(cuda)).sum().backward()
    compare_all(dict(m.vision_encoder.visual.named_parameters()), ref_grads, TOL, f"vit {tag}")


def test_bert_full_gradients(setup, cuda):
    vtype, tag, m, sd = setup
    if tag != "b16_d2":
        pytest.skip("tower independent")
    fx = golden("bert.pt")
    g = torch.Generator().manual_seed(fx["meta"]["seed"])
    b, S, E = fx["meta"]["b"], fx["meta"]["S"], fx["meta"]["E"]
    ids = torch.randint(1000, 30000, (b, S), generator=g)
    ids[:, 0] = 101
    mask = (torch.arange(S)[None] < fx["meta"]["lens"][:, None]).long()
    ids = ids * mask
    cond = torch.randn((b, E, 768), generator=g)
    m3 = torch.tril(mask.unsqueeze(1).expand(-1, S, -1)).contiguous()
    sdo = oracle_sd(sd)
    cr_o = cond.clone().requires_grad_(True)
    ro = O.bert_mlm(sdo, ids, m3, cr_o, fx["labels"])
    assert abs(ro["loss"].item() - fx["causal_loss"].item()) < 1e-5 * fx["causal_loss"].item()
    ro["loss"].backward()
    pre = "multimodal_encoder."
    for n, d in fx["causal_grads"].items():
        assert grad_digest_check(d, sdo[pre + n].grad, None) < 1e-4, ("oracle vs reference digest", n)
    # every BERT / LM-head parameter the causal cross-attention pass touches (the decoder weight is the tied word-embedding table)
    ref_grads = {k[len(pre):]: v.grad for k, v in sdo.items()
                 if k.startswith(pre) and v.requires_grad and v.grad is not None and not k.endswith("cls.predictions.decoder.weight")}
    assert len(ref_grads) > 150
    me = m.multimodal_encoder
    m.zero_grad(set_to_none=True)
    cr = cond.to(cuda).clone().requires_grad_(True)
    with runtime.precision(torch.float16):
        o = me(input_ids=ids.to(cuda), attention_mask=m3.to(cuda), encoder_hidden_states=cr, labels=fx["labels"].to(cuda))
        o.loss.backward()
    named = dict(me.named_parameters())
    named.pop("cls.predictions.decoder.weight", None)
    compare_all(named, ref_grads, TOL, "bert causal x-attn + LM head")
    assert full_err(cr.grad, cr_o.grad) < TOL


def test_alignment_step_full_gradients(setup, cuda):
    """ITC + ITM + CAP with injected negatives / token masks: every parameter of the model that receives a gradient."""
    vtype, tag, m, sd = setup
    fx = golden(f"loss_{tag}.pt")
    r = fx["W1"]
    b = fx["meta"]["b"]
    inp = synth_inputs(dict(b=b, vision=2, audio=1, S=12), seed=1234)
    injected = {st: {k: r["inj"][st][k] for k in ("neg_cond_idx", "neg_text_idx")} for st in ("tva", "tv")}
    injected["cap"] = r["inj"]["cap"]
    sdo = oracle_sd(sd)
    ref, _ = O.mico_forward(sdo, O.ARCHS[vtype], inp, fx["meta"]["task"], dict(itm_ratio=0.1), injected=injected)
    for k, v in r["losses"].items():   # the oracle against the reference-generated losses
        assert abs(float(ref[k].detach()) - float(v)) <= 1e-4 * max(abs(float(v)), 1e-6), (k, float(ref[k].detach()), float(v))
    sum(ref.values()).backward()
    for n, d in r["grads"].items():
        assert grad_digest_check(d, sdo[n].grad, None) < 1e-3, ("oracle vs reference digest", n)
    ref_grads = {k: v.grad for k, v in sdo.items() if v.requires_grad and v.grad is not None and not k.endswith("cls.predictions.decoder.weight")}
    batch = {k: v.to(cuda) for k, v in inp.items()}
    batch["_injected"] = injected
    m.zero_grad(set_to_none=True)
    with runtime.precision(torch.float16):
        out = m(batch, fx["meta"]["task"], compute_loss=True)
        sum(out.values()).backward()
    named = {n: p for n, p in m.named_parameters() if not n.endswith("cls.predictions.decoder.weight")}
    # parameters the oracle gives an exactly-zero gradient (unused heads reached through a zero path) are compared in absolute terms
    nz = {n: gr for n, gr in ref_grads.items() if gr.abs().max() > 0}
    compare_all(named, nz, 5e-2, f"alignment step {tag}")
    for n, gr in ref_grads.items():
        if n not in nz and named[n].grad is not None:
            assert named[n].grad.abs().max().item() == 0.0, n


@pytest.mark.parametrize("pc", ["parity", "timed"])
def test_vit_g_depth40_full_gradients(cuda, pc):
    """The full 40-block EVA01-g/14 tower, one image: every parameter gradient element-wise against the CPU oracle, in the parity
    configuration and in the precision bench.py times (16-bit stored gradients compound over depth; the depth-2
    checks above cannot see that).  The oracle's forward is re-pinned to the reference's own full-depth output inside the test."""
    from common import precision_config, rel_err
    torch.set_num_threads(min(32, torch.get_num_threads() or 32))
    fx = golden("vit_g14_full.pt")
    m, sd = build_model("evaclip01_giant", None, device=cuda)
    g = torch.Generator().manual_seed(fx["meta"]["input_seed"])
    x = torch.randn((1, 3, 224, 224), generator=g)
    pre = "vision_encoder.visual."
    sdo = {k: (v.clone().requires_grad_(True) if k.startswith(pre) and v.is_floating_point() else v) for k, v in sd.items()}
    ref = O.eva_vit_forward(sdo, x, O.ARCHS["evaclip01_giant"])
    assert ((ref[0, [0, 1, 128, 256]].detach() - fx["rows"]).abs().max() / fx["amax"]).item() < 1e-5      # oracle == reference at depth 40
    w = torch.randn(ref.shape, generator=torch.Generator().manual_seed(77)) / ref.numel() ** 0.5
    (ref * w).sum().backward()
    ref_grads = {k[len(pre):]: v.grad for k, v in sdo.items() if k.startswith(pre) and torch.is_tensor(v) and v.requires_grad and v.grad is not None}
    assert len(ref_grads) > 400
    m.zero_grad(set_to_none=True)
    with precision_config(pc):
        out = m.vision_encoder.visual(x.to(cuda), return_all_features=True)
        e_fwd = rel_err(out, ref)
        (out * w.to(cuda)).sum().backward()
    print(f"depth-40 g/14 [{pc}] forward {e_fwd:.2e}")
    assert e_fwd < 1e-3
    errs = compare_all(dict(m.vision_encoder.visual.named_parameters()), ref_grads, TOL, f"vit g/14 depth 40 [{pc}]")
    by_block = {}
    for n, e in errs.items():
        if n.startswith("blocks."):
            by_block.setdefault(int(n.split(".")[1]), []).append(e)
    print("worst per block:", " ".join(f"{b}:{max(v):.1e}" for b, v in sorted(by_block.items())))
