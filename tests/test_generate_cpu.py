"""Host-side pieces of caption decoding (no GPU): the causal mask growth of bert.py:1110-1117, the n-best bookkeeping with
length penalty, and the oracle search itself (num_beams = 1 equals the stepwise argmax chain; finished hypotheses win over
longer ones exactly when sum_logprob / len**0.6 says so)."""
import torch

from common import build_model
from oracle import mico_oracle as O


def test_mask_growth_and_nbest():
    from mico_amd.model.bert import BertForMaskedLM, _BeamHypotheses
    m = torch.ones(2, 1, 1, dtype=torch.long)
    for n in range(2, 6):
        m = BertForMaskedLM.update_attention_mask(m)
        assert m.shape == (2, n, n) and torch.equal(m[0], torch.tril(torch.ones(n, n, dtype=torch.long)))
        assert torch.equal(m, O.grow_mask(torch.tril(torch.ones(2, n - 1, n - 1, dtype=torch.long))))
    h = _BeamHypotheses(2, 0.6)
    assert not h.is_done(-1.0, 3)
    h.add(torch.tensor([101, 5]), -4.0)             # score -4 / 2**0.6
    h.add(torch.tensor([101, 5, 6, 7]), -6.0)       # score -6 / 4**0.6
    assert abs(h.worst_score - (-4.0 / 2 ** 0.6)) < 1e-6
    h.add(torch.tensor([101]), -1.0)                # better than both: evicts the worst
    assert len(h.beams) == 2 and abs(h.worst_score - (-6.0 / 4 ** 0.6)) < 1e-6
    h.add(torch.tensor([101, 9, 9]), -30.0)         # worse than the worst: ignored
    assert len(h.beams) == 2 and sorted(len(b[1]) for b in h.beams) == [1, 4]
    assert h.is_done(-20.0, 5) and not h.is_done(-1.0, 5)


def test_oracle_greedy_is_argmax_chain():
    torch.set_num_threads(8)
    _, sd = build_model("evaclip02_base", 1)
    sd = dict(sd)
    sd["multimodal_encoder.cls.predictions.decoder.weight"] = sd["multimodal_encoder.bert.embeddings.word_embeddings.weight"]
    cond = torch.randn(2, 5, 768, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        out = O.generate_beam(sd, cond, 4, 1, 0.6)
        ids, mask = torch.full((2, 1), 101), torch.ones(2, 1, 1, dtype=torch.long)
        for _ in range(4):
            ids = torch.cat([ids, O.decode_step_logits(sd, ids, mask, cond).argmax(-1)[:, None]], 1)
            mask = O.grow_mask(mask)
    assert out.tolist() == ids.tolist()
