"""Edge cases through the C-ABI and the module surface: degenerate sizes (one row / one query / one frame / batch of one), a
text that is only [CLS][SEP] + padding, zero-row launches (no-ops), and loud failures for what is unsupported."""
import pytest
import torch
import torch.nn.functional as F

from common import build_model, rel_err
from mico_amd import runtime
from mico_amd.weights import synth_inputs
from oracle import mico_oracle as O

pytestmark = pytest.mark.gpu


def test_degenerate_kernel_sizes(cuda):
    from mico_amd import ops
    from mico_amd._lib import MicoHipError
    torch.manual_seed(0)
    # GEMM far below one tile, and exactly one row
    for M, N, K in ((1, 8, 8), (3, 264, 40), (130, 4, 16)):
        a = torch.randn(M, K, device=cuda).bfloat16()
        w = torch.randn(N, K, device=cuda).bfloat16()
        out = torch.empty(M, N, device=cuda)
        ops.gemm(a, w, out)
        assert rel_err(out, a.float() @ w.float().t()) < 1e-5
    # LayerNorm on one row; zero rows is a no-op
    x = torch.randn(1, 768, device=cuda)
    g, b = torch.randn(768, device=cuda), torch.randn(768, device=cuda)
    y = torch.empty(1, 768, device=cuda)
    ops.layernorm_fwd(x, g, b, 1e-12, out32=y, dtype=torch.bfloat16)
    assert rel_err(y, F.layer_norm(x, (768,), g, b, 1e-12)) < 1e-5
    ops.layernorm_fwd(x[:0], g, b, 1e-12, out32=y[:0], dtype=torch.bfloat16)
    # attention with a single query and a single key (the softmax of one score is 1: output = v)
    qkv = torch.randn(2, 1, 3 * 128, device=cuda).half()
    o = torch.empty(2, 1, 128, device=cuda, dtype=torch.float16)
    lse = torch.empty(2, 2, 1, device=cuda)
    ops.attn_fwd(qkv.view(2, 384), qkv.view(2, 384)[:, 128:], qkv.view(2, 384)[:, 256:], o.view(2, 128), lse, B=2, H=2, Sq=1, Sk=1,
                 hd=64, scale=0.125, q_strides=(384, 384), k_strides=(384, 384), v_strides=(384, 384), o_strides=(128, 128))
    assert torch.equal(o.view(2, 128), qkv.view(2, 384)[:, 256:])
    # unsupported shapes fail loudly with the library's message
    with pytest.raises(MicoHipError, match="multiples of 8"):
        ops.gemm(torch.zeros(4, 12, device=cuda).bfloat16()[:, :10], torch.zeros(8, 10, device=cuda).bfloat16(), torch.empty(4, 8, device=cuda))


def test_batch_of_one_and_degenerate_text(cuda):
    """b = 1, one frame per modality, and a caption that is only [CLS][SEP] (+ padding): features against the oracle, and a full
    alignment step whose in-batch negative sampling has nothing to sample from (the own-sample weight is zeroed: the draw falls on
    the 1e-4 floor) still returns finite losses."""
    torch.set_num_threads(16)
    m, sd = build_model("evaclip02_base", 2, device=cuda)
    sdo = dict(sd)
    sdo["multimodal_encoder.cls.predictions.decoder.weight"] = sdo["multimodal_encoder.bert.embeddings.word_embeddings.weight"]
    inp = synth_inputs(dict(b=1, vision=1, audio=1, depth=1, S=9), seed=3)
    inp["input_ids"] = torch.tensor([[101, 102, 0, 0, 0, 0, 0, 0, 0]])
    inp["attention_mask"] = torch.tensor([[1, 1, 0, 0, 0, 0, 0, 0, 0]])
    dev = {k: v.to(cuda) for k, v in inp.items()}
    with torch.no_grad():
        ref = O.encode_batch(sdo, O.ARCHS["evaclip02_base"], inp)
    with runtime.precision(torch.float16), torch.no_grad():
        enc = m.encode_batch(dict(dev))
        assert rel_err(enc["feat_t"], ref["feat_t"]) < 1e-3
        for c in ("v", "a", "d", "va", "vd"):
            assert rel_err(m._feat_cond(enc, c), O.feat_cond(sdo, ref, c)) < 1e-3, c
    inp2 = synth_inputs(dict(b=2, vision=1, audio=1, S=9), seed=4)
    inp2["input_ids"][1] = torch.tensor([101, 102, 0, 0, 0, 0, 0, 0, 0])
    inp2["attention_mask"][1] = torch.tensor([1, 1, 0, 0, 0, 0, 0, 0, 0])
    m.train()
    with runtime.precision(torch.bfloat16):
        out = m({k: v.to(cuda) for k, v in inp2.items()}, "ret%tva%tv_cap%tva")
        assert all(torch.isfinite(v) for v in out.values())
        sum(out.values()).backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
    with pytest.raises(AssertionError):
        m({k: v.to(cuda) for k, v in inp2.items()}, "ret%tq")            # unknown sub-task
    with pytest.raises(NotImplementedError):
        m({k: v.to(cuda) for k, v in inp2.items()}, "qa%tv")              # the QA task family is not part of this path
