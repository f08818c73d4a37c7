"""Caption decoding: BertForMaskedLM.generate on the device against the CPU restatement
oracle.mico_oracle.generate_beam - token ids bit-exact.  The [SEP] output bias is raised so that end-of-sequence
candidates actually occur (finished-hypothesis bookkeeping, length penalty, early close, eos/pad fill); num_beams = 1 must
equal the step-by-step argmax chain."""
import pytest
import torch

from common import build_model
from mico_amd import runtime
from oracle import mico_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("num_beams,sep_bias,max_new", [(1, 0.0, 6), (3, 0.0, 6), (3, 1.2, 8), (3, 1.5, 8), (3, 1.7, 8), (2, 1.6, 8), (2, 1.8, 8)])
def test_generate_matches_oracle(cuda, num_beams, sep_bias, max_new):
    torch.set_num_threads(16)
    m, sd = build_model("evaclip02_base", 1, device=cuda)
    sdo = dict(sd)
    sdo["multimodal_encoder.cls.predictions.decoder.weight"] = sdo["multimodal_encoder.bert.embeddings.word_embeddings.weight"]
    bias = sdo["multimodal_encoder.cls.predictions.bias"].clone()
    bias[102] += sep_bias
    sdo["multimodal_encoder.cls.predictions.bias"] = bias
    with torch.no_grad():
        m.multimodal_encoder.cls.predictions.bias.copy_(bias.to(cuda))
    g = torch.Generator().manual_seed(3)
    cond = torch.randn(3, 7, 768, generator=g)
    with torch.no_grad():
        ref = O.generate_beam(sdo, cond, max_new, num_beams, 0.6)
    tk = m.multimodal_encoder.tokenizer
    with runtime.precision(torch.float16):
        init = torch.full((3, 1), tk.bos_token_id, dtype=torch.long, device=cuda)
        out = m.multimodal_encoder.generate(input_ids=init, attention_mask=init.new_ones(3, 1, 1), encoder_hidden_states=cond.to(cuda),
                                            max_new_tokens=max_new, num_beams=num_beams, eos_token_id=tk.sep_token_id,
                                            pad_token_id=tk.pad_token_id, length_penalty=0.6)
    print(num_beams, sep_bias, out.tolist(), ref.tolist())
    assert out.cpu().tolist() == ref.tolist()
    if sep_bias > 0:
        assert (ref == 102).any(), "the case was meant to produce finished hypotheses"
    if num_beams == 1 and sep_bias == 0:
        ids = torch.full((3, 1), 101)
        mask = torch.ones(3, 1, 1, dtype=torch.long)
        with torch.no_grad():
            for _ in range(max_new):
                t = O.decode_step_logits(sdo, ids, mask, cond).argmax(-1)
                ids = torch.cat([ids, t[:, None]], 1)
                mask = O.grow_mask(mask)
        assert ids.tolist() == ref.tolist()
    with pytest.raises(TypeError):
        m.multimodal_encoder.generate(input_ids=init, attention_mask=init.new_ones(3, 1, 1), temperature=0.7)


@pytest.mark.parametrize("sep_bias,max_new", [(0.0, 6), (2.5, 8), (4.0, 8)])
def test_generate_top_k_sampling_matches_oracle(cuda, sep_bias, max_new):
    """captioner_mode decode (vast.py:526-536: do_sample=True, top_k=10) with injected uniform numbers: token ids bit-exact against
    oracle.generate_sample, incl. rows that finish early (eos, then pad) when the [SEP] bias is raised."""
    torch.set_num_threads(16)
    m, sd = build_model("evaclip02_base", 1, device=cuda)
    sdo = dict(sd)
    sdo["multimodal_encoder.cls.predictions.decoder.weight"] = sdo["multimodal_encoder.bert.embeddings.word_embeddings.weight"]
    bias = sdo["multimodal_encoder.cls.predictions.bias"].clone()
    bias[102] += sep_bias
    sdo["multimodal_encoder.cls.predictions.bias"] = bias
    with torch.no_grad():
        m.multimodal_encoder.cls.predictions.bias.copy_(bias.to(cuda))
    g = torch.Generator().manual_seed(4)
    cond = torch.randn(4, 7, 768, generator=g)
    noise = torch.rand(4, max_new, generator=g)
    tk = m.multimodal_encoder.tokenizer
    me = m.multimodal_encoder
    with runtime.precision(torch.float16), torch.no_grad():
        init = torch.full((4, 1), tk.bos_token_id, dtype=torch.long, device=cuda)
        out = me.generate(input_ids=init, attention_mask=init.new_ones(4, 1, 1), encoder_hidden_states=cond.to(cuda),
                          max_new_tokens=max_new, do_sample=True, top_k=10, eos_token_id=tk.sep_token_id,
                          pad_token_id=tk.pad_token_id, sample_noise=noise)
        # the oracle's sampling loop over the PRODUCT's step logits: the search (top-k, softmax, inverse-CDF draw, eos / pad
        # bookkeeping, stop rule) must agree token for token ...
        step = lambda ids, mask: me.next_token_logits(ids.to(cuda), mask.to(cuda), cond.to(cuda), None).float().cpu()
        ref = O.generate_sample(sdo, cond, max_new, 10, noise, step_logits=step)
        # (the logits themselves are gated in test_model_gpu.py::test_bert; with random-init weights the ten largest of 30522 nearly
        # flat logits reorder within 16-bit rounding, so the fp32 oracle's own chain is not a bit-exact target here)
    print(sep_bias, out.tolist(), ref.tolist())
    assert out.cpu().tolist() == ref.tolist()
    for row in ref.tolist():       # a finished row: eos once, pad from then on (whenever the raised [SEP] bias makes it happen)
        if 102 in row:
            assert all(t == 0 for t in row[row.index(102) + 1:]), row


def test_forward_cap_captioner_mode(cuda):
    """MiCo.forward(batch, "cap%tv", compute_loss=False) with config.captioner_mode: generate_nums sampled captions per sample,
    sample-major (vast.py:519-536), same ids as the oracle under the same injected noise."""
    from mico_amd.weights import synth_inputs
    torch.set_num_threads(16)
    m, sd = build_model("evaclip02_base", 1, device=cuda, max_caption_len=5, captioner_mode=True, generate_nums=2)
    sdo = dict(sd)
    sdo["multimodal_encoder.cls.predictions.decoder.weight"] = sdo["multimodal_encoder.bert.embeddings.word_embeddings.weight"]
    inp = synth_inputs(dict(b=2, vision=2, S=8), seed=8)
    noise = torch.rand(4, 5, generator=torch.Generator().manual_seed(1))
    batch = {k: v.to(cuda) for k, v in inp.items()}
    batch["_injected"] = {"sample_noise": noise}
    me, tk = m.multimodal_encoder, m.multimodal_encoder.tokenizer
    with runtime.precision(torch.float16), torch.no_grad():
        out = m(batch, "cap%tv", compute_loss=False)
        # the same decode by hand: condition tokens repeated generate_nums times sample-major, top-k 10 sampling with the same noise
        cond = m._condition_feats(m.encode_batch(dict(batch)), "v").repeat_interleave(2, dim=0).contiguous()
        step = lambda ids, mask: me.next_token_logits(ids.to(cuda), mask.to(cuda), cond, None).float().cpu()
        ref = O.generate_sample(sdo, cond.float().cpu(), 5, 10, noise, step_logits=step)
    want = tk.batch_decode(ref[:, 1:], skip_special_tokens=True)
    assert out == {"generated_captions_tv": want} and len(want) == 4


def test_forward_cap_evaluation_dict(cuda):
    """MiCo.forward(batch, "cap%tv", compute_loss=False) -> {"generated_captions_tv": [str] * b} (vast.py:513-547), same ids as the
    oracle's beam search on the oracle's condition tensor."""
    from mico_amd.weights import synth_inputs
    torch.set_num_threads(16)
    m, sd = build_model("evaclip02_base", 1, device=cuda, max_caption_len=6)
    sdo = dict(sd)
    sdo["multimodal_encoder.cls.predictions.decoder.weight"] = sdo["multimodal_encoder.bert.embeddings.word_embeddings.weight"]
    inp = synth_inputs(dict(b=2, vision=2, S=8), seed=8)
    with torch.no_grad():
        enc = O.encode_batch(sdo, O.ARCHS["evaclip02_base"], inp)
        ref = O.generate_beam(sdo, O.condition_feats(enc, "v"), 6, 3, 0.6)
    with runtime.precision(torch.float16), torch.no_grad():
        out = m({k: v.to(cuda) for k, v in inp.items()}, "cap%tv", compute_loss=False)
    want = m.multimodal_encoder.tokenizer.batch_decode(ref[:, 1:], skip_special_tokens=True)
    assert out == {"generated_captions_tv": want}
