"""Train-mode BERT dropout (bert.py:148,267,295,373; p = 0.1 hidden and attention-probability).  The product never stores a
mask: every kernel derives keep/drop from a counter hash of (seed, site, element index); the oracle restates the hash
(drop_mask) and applies the reference's x * mask / (1 - p).  Forward values and gradients must agree, for self-attention only,
2-D-mask cross attention and the causal 3-D mask with labels; and the mask statistics must be Bernoulli(1 - p)."""
import pytest
import torch

from common import build_model, rel_err
from mico_amd import ops, runtime
from oracle import mico_oracle as O

pytestmark = pytest.mark.gpu


def test_dropout_kernel_matches_restated_hash(cuda):
    x = torch.ones(37, 768, device=cuda)
    ops.dropout_(x, (0.1, 1234567, 19))
    ref = O.drop_mask(1234567, 19, (37, 768), 0.1)
    assert torch.equal(x.cpu(), ref)
    frac = (x == 0).float().mean().item()
    assert 0.08 < frac < 0.12
    h = torch.ones(64, 96, device=cuda, dtype=torch.bfloat16)
    ops.dropout_(h, (0.5, 7, 100000))
    assert torch.equal((h != 0).cpu(), O.drop_mask(7, 100000, (64, 96), 0.5) != 0)


@pytest.mark.parametrize("mode", ["self", "cross", "causal_cross"])
def test_bert_dropout_parity(cuda, mode):
    torch.set_num_threads(16)
    m, sd = build_model("evaclip02_base", 1, device=cuda)
    bert = m.multimodal_encoder.bert
    g = torch.Generator().manual_seed(4)
    b, S, E = 3, 10, 9
    ids = torch.randint(1000, 30000, (b, S), generator=g)
    am = torch.ones(b, S, dtype=torch.long)
    am[1, 7:] = 0
    ids[1, 7:] = 0
    cond = torch.randn(b, E, 768, generator=g) if mode != "self" else None
    mask = torch.tril(am.unsqueeze(1).expand(-1, S, -1).clone()) if mode == "causal_cross" else am
    names = ["embeddings.word_embeddings.weight", "encoder.layer.3.attention.self.value.weight", "encoder.layer.0.output.dense.weight",
             "encoder.layer.5.attention.output.LayerNorm.weight", "encoder.layer.11.intermediate.dense.bias"]
    if cond is not None:
        names += ["encoder.layer.2.crossattention.self.key.weight", "encoder.layer.7.crossattention.output.dense.weight"]
    pre = "multimodal_encoder.bert."
    sdo = {k: (v.clone().requires_grad_(True) if k[len(pre):] in names and k.startswith(pre) else v) for k, v in sd.items()}
    cond_o = cond.clone().requires_grad_(True) if cond is not None else None
    w = torch.randn(b, S, 768, generator=g) / (b * S * 768) ** 0.5
    with O.bert_dropout(0.1, 0.1, [4242]):
        ref = O.bert_forward(sdo, ids, mask, cond_o)
    (ref * w).sum().backward()
    bert.train()
    bert.dropout_seed_source = lambda: 4242
    cond_d = cond.to(cuda).requires_grad_(True) if cond is not None else None
    try:
        with runtime.precision(torch.float16):
            m.zero_grad(set_to_none=True)
            out = bert(ids.to(cuda), mask.to(cuda), cond_d).last_hidden_state
            (out * w.to(cuda)).sum().backward()
    finally:
        bert.dropout_seed_source = None
        bert.eval()
    e = rel_err(out, ref)
    print(mode, "sequence output", f"{e:.2e}")
    assert e < 1e-3
    named = dict(bert.named_parameters())
    for n in names:
        ge = rel_err(named[n].grad, sdo[pre + n].grad)
        print("  ", n, f"{ge:.2e}")
        assert ge < 2e-2, (n, ge)
    if cond is not None:
        ge = rel_err(cond_d.grad, cond_o.grad)
        print("   cond", f"{ge:.2e}")
        assert ge < 2e-2
    # eval mode must be unaffected by the seed source
    with runtime.precision(torch.float16), torch.no_grad():
        ev = bert(ids.to(cuda), mask.to(cuda), cond_d).last_hidden_state
        ref_ev = O.bert_forward(sd, ids, mask, cond)
    assert rel_err(ev, ref_ev) < 1e-3
