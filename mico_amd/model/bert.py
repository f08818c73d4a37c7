"""Host-side mirror of the reference's BERT-with-cross-attention text head (model/bert.py:81-1108): same module tree and
parameter names (`bert.embeddings.*`, `bert.encoder.layer.N.*`, `cls.predictions.*`), same call surface
(BertForMaskedLM(input_ids, attention_mask[2-D|3-D], encoder_hidden_states, labels) -> obj with .loss / .logits /
.sequence_output), arithmetic in mico_amd.functional.BertFn / LMHeadLossFn on libmico_hip.so.

BERT_CONFIG restates model/bert-base-uncased-crossattn/config.json (shape contract).
"""
import os

import torch
from torch import nn

from .. import functional as Fn

BERT_CONFIG = dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                   max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, pad_token_id=0,
                   hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, add_cross_attention=True, is_decoder=True)

TOKENIZER_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tokenizer")


class _Embeddings(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.word_embeddings = nn.Embedding(c["vocab_size"], c["hidden_size"], padding_idx=c["pad_token_id"])
        self.position_embeddings = nn.Embedding(c["max_position_embeddings"], c["hidden_size"])
        self.token_type_embeddings = nn.Embedding(c["type_vocab_size"], c["hidden_size"])
        self.LayerNorm = nn.LayerNorm(c["hidden_size"], eps=c["layer_norm_eps"])
        self.register_buffer("position_ids", torch.arange(c["max_position_embeddings"]).expand((1, -1)))


class _SelfAttention(nn.Module):
    def __init__(self, c):
        super().__init__()
        h = c["hidden_size"]
        self.query, self.key, self.value = nn.Linear(h, h), nn.Linear(h, h), nn.Linear(h, h)


class _SelfOutput(nn.Module):
    def __init__(self, c, in_features=None):
        super().__init__()
        self.dense = nn.Linear(in_features or c["hidden_size"], c["hidden_size"])
        self.LayerNorm = nn.LayerNorm(c["hidden_size"], eps=c["layer_norm_eps"])


class _Attention(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.self = _SelfAttention(c)
        self.output = _SelfOutput(c)


class _Intermediate(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c["hidden_size"], c["intermediate_size"])


class _Layer(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.attention = _Attention(c)
        self.crossattention = _Attention(c)
        self.intermediate = _Intermediate(c)
        self.output = _SelfOutput(c, c["intermediate_size"])


class _Encoder(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.layer = nn.ModuleList([_Layer(c) for _ in range(c["num_hidden_layers"])])


class _Out(dict):
    """attr-dict like the reference's easydict return value (bert.py:1093-1097)."""
    __getattr__ = dict.get

    def __setattr__(self, k, v):
        self[k] = v


def extended_attention_mask(attention_mask):
    """bert.py:697-781: 2-D [b,S] key mask or 3-D [b,S,S] -> additive fp32 (1 - m) * -10000 (no automatic causal mask)."""
    if attention_mask.dim() not in (2, 3):
        raise ValueError(f"Wrong shape for attention_mask (shape {tuple(attention_mask.shape)})")
    return ((1.0 - attention_mask.to(torch.float32)) * -10000.0).contiguous()


class BertModel(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.config = c
        self.embeddings = _Embeddings(c)
        self.encoder = _Encoder(c)
        self._spec = None
        self.dropout_seed_source = None   # callable -> int; None: one draw per pass from torch's host generator

    def _bert_spec(self):
        named = list(self.named_parameters())
        names = [n for n, _ in named]
        if self._spec is None or self._spec.names != names:
            self._spec = Fn.BertSpec(names, len(self.encoder.layer), self.config["num_attention_heads"],
                                     self.config["hidden_size"], self.config["intermediate_size"], self.config["layer_norm_eps"])
        return self._spec, [p for _, p in named]

    def project_cross_kv(self, cond_own, cond_neg=None):
        """Cross-attention K/V memory of condition tokens for all layers, to be shared by several passes of one training step
        (`cross_kv=` of forward): (kv_own, kv_neg) for the batch's own tokens [b, E, D] and, optionally, ITM hard negatives.
        Differentiable with respect to the tokens and the key / value projections (functional.CrossKVFn)."""
        spec, params = self._bert_spec()
        kvp = []
        for li in range(spec.L):
            ca = f"encoder.layer.{li}.crossattention.self."
            kvp += [params[spec.idx[ca + n]] for n in ("key.weight", "key.bias", "value.weight", "value.bias")]
        session = Fn.DkvSession()
        kv_own, kv_neg = Fn.CrossKVFn.apply(spec, session, cond_own, cond_neg, *kvp)
        if kv_own.requires_grad:
            # the passes that read this memory find the session on the tensor (BertFn.forward) and keep ONE gradient buffer for the own set
            kv_own._mico_dkv = session
        return kv_own, kv_neg

    def forward(self, input_ids=None, attention_mask=None, encoder_hidden_states=None, kv_cache=None, cross_kv=None, **_):
        """kv_cache (dict, inference only): holds the cross-attention K/V projections of `encoder_hidden_states` across calls -
        the caller guarantees the condition tokens do not change between the calls that share the dict.
        cross_kv (training): (kv_own, kv_neg) from project_cross_kv instead of encoder_hidden_states; a batch of b entries attends
        to kv_own, a batch of 3 b entries is the ITM triplet [own | hard negative | own] and needs kv_neg as well."""
        if input_ids is None:
            raise ValueError("You have to specify input_ids")
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        spec, params = self._bert_spec()
        drop = None
        ph, pa = self.config["hidden_dropout_prob"], self.config["attention_probs_dropout_prob"]
        if self.training and (ph > 0 or pa > 0):
            # nn.Dropout sites of bert.py:148,267,295,373: masks come from a counter hash of (seed, site, element), one seed per pass
            seed = self.dropout_seed_source() if self.dropout_seed_source is not None else int(torch.randint(0, 2 ** 31 - 1, (1,)))
            drop = (float(ph), float(pa), int(seed))
        if kv_cache is not None:
            if torch.is_grad_enabled() and any(p.requires_grad for p in params):
                raise RuntimeError("kv_cache is an inference feature: call under torch.no_grad()")
            drop = {"kv_cache": kv_cache}
        kv_own, kv_neg = (None, None)
        if cross_kv is not None:
            if encoder_hidden_states is not None or kv_cache is not None:
                raise ValueError("cross_kv replaces encoder_hidden_states / kv_cache")
            kv_own, kv_neg = cross_kv
            if kv_neg is not None and input_ids.shape[0] % 3:
                raise ValueError("cross_kv with hard negatives expects the ITM triplet batch [own | negative | own]")
        seq = Fn.BertFn.apply(spec, input_ids, extended_attention_mask(attention_mask), encoder_hidden_states, drop, kv_own, kv_neg,
                              *params)
        return _Out(last_hidden_state=seq)


class _Transform(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c["hidden_size"], c["hidden_size"])
        self.LayerNorm = nn.LayerNorm(c["hidden_size"], eps=c["layer_norm_eps"])


class _Predictions(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.transform = _Transform(c)
        self.decoder = nn.Linear(c["hidden_size"], c["vocab_size"], bias=False)
        self.bias = nn.Parameter(torch.zeros(c["vocab_size"]))
        self.decoder.bias = self.bias   # bert.py:604


class _MLMHead(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.predictions = _Predictions(c)


class _LazyLogits:
    def __init__(self, fn):
        self._fn, self._v = fn, None

    def get(self):
        if self._v is None:
            self._v = self._fn()
        return self._v


class _MLMOut(_Out):
    """`.logits` is computed on first access: the reference always evaluates the 768x30522 LM head (bert.py:1085) even when
    only .sequence_output is consumed; results are identical, the wasted GEMM is not."""

    def __getattr__(self, k):
        if k == "logits":
            lazy = dict.get(self, "_lazy_logits")
            return lazy.get() if lazy is not None else None
        return dict.get(self, k)


class BertForMaskedLM(nn.Module):
    def __init__(self, config=None):
        super().__init__()
        c = dict(BERT_CONFIG)
        if config:
            c.update(config)
        self.config = c
        self.bert = BertModel(c)
        self.cls = _MLMHead(c)
        self._init_weights()
        # weight tying (transformers==4.31 post_init via get_output_embeddings, bert.py:1038-1041)
        self.cls.predictions.decoder.weight = self.bert.embeddings.word_embeddings.weight
        self.tokenizer = None

    def _init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Linear):
                m.weight.data.normal_(mean=0.0, std=0.02)
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, nn.Embedding):
                m.weight.data.normal_(mean=0.0, std=0.02)
                if m.padding_idx is not None:
                    m.weight.data[m.padding_idx].zero_()
            elif isinstance(m, nn.LayerNorm):
                m.bias.data.zero_()
                m.weight.data.fill_(1.0)

    def get_output_embeddings(self):
        return self.cls.predictions.decoder

    def _head_params(self):
        pr = self.cls.predictions
        return (pr.transform.dense.weight, pr.transform.dense.bias, pr.transform.LayerNorm.weight, pr.transform.LayerNorm.bias,
                pr.decoder.weight, pr.bias)

    def forward(self, input_ids=None, attention_mask=None, encoder_hidden_states=None, labels=None, cross_kv=None, **_):
        seq = self.bert(input_ids, attention_mask, encoder_hidden_states, cross_kv=cross_kv).last_hidden_state
        out = _MLMOut(loss=None, sequence_output=seq)
        hp = self._head_params()
        if labels is not None:
            out["loss"] = Fn.LMHeadLossFn.apply(seq, labels, *hp)
        dict.__setitem__(out, "_lazy_logits", _LazyLogits(lambda: Fn.LMLogitsFn.apply(seq.detach(), *[p.detach() for p in hp])))
        return out

    # ---- caption decoding (inference_demo.py:161-171; the [MASK]-append protocol of bert.py:1110-1143) -------------------------
    @staticmethod
    def update_attention_mask(attention_mask):
        """bert.py:1110-1117: grow a [b,n,n] mask to [b,n+1,n+1]; the new row copies the last row and sees itself."""
        b, n, _ = attention_mask.shape
        up = attention_mask.new_zeros(b, n + 1, n + 1)
        up[:, :n, :n] = attention_mask
        up[:, n, :n] = attention_mask[:, n - 1, :n]
        up[:, n, n] = 1
        return up

    def prepare_inputs_for_generation(self, input_ids, attention_mask=None, encoder_hidden_states=None, **_):
        """bert.py:1126-1143: append one [MASK] token whose output row predicts the next token."""
        dummy = torch.full((input_ids.shape[0], 1), self.tokenizer.mask_token_id, dtype=torch.long, device=input_ids.device)
        return {"input_ids": torch.cat([input_ids, dummy], dim=1), "attention_mask": self.update_attention_mask(attention_mask),
                "encoder_hidden_states": encoder_hidden_states}

    @torch.no_grad()
    def next_token_logits(self, input_ids, attention_mask, encoder_hidden_states, kv_cache=None):
        """One decode step of the reference protocol: logits [rows, vocab] of the appended [MASK] position.  Only that row
        goes through the 768x30522 LM head (the reference evaluates it for every position and slices, bert.py:1085); with a
        kv_cache dict the cross-attention K/V of the condition tokens are projected once per decode, not once per step."""
        inp = self.prepare_inputs_for_generation(input_ids, attention_mask, encoder_hidden_states)
        seq = self.bert(inp["input_ids"], inp["attention_mask"], inp["encoder_hidden_states"], kv_cache=kv_cache).last_hidden_state
        last = seq[:, -1:, :].contiguous()
        return Fn.LMLogitsFn.apply(last, *[p.detach() for p in self._head_params()])[:, 0, :]

    @torch.no_grad()
    def generate(self, input_ids=None, attention_mask=None, encoder_hidden_states=None, max_new_tokens=20, num_beams=1,
                 eos_token_id=None, pad_token_id=None, length_penalty=1.0, do_sample=False, top_k=50, sample_noise=None, **unused):
        """Beam search with the semantics of transformers==4.31 GenerationMixin.generate / BeamSearchScorer as the reference
        calls it (inference_demo.py:164-171: num_beams 3, length_penalty 0.6, early_stopping False, no logits processors):
        2*num_beams candidates per step, finished hypotheses scored sum_logprob / len**length_penalty, the "cannot improve"
        stop heuristic, finalisation with the open beams, eos-terminated pad-filled output.  The search bookkeeping runs on
        the host over 2*num_beams candidates per sample; the model step and log-softmax / top-k run on the device."""
        if unused:
            raise TypeError(f"generate(): unsupported arguments {sorted(unused)}")
        if do_sample:
            if int(num_beams) != 1:
                raise TypeError("generate(): do_sample with num_beams > 1 (beam sampling) is not used by the reference and not provided")
            return self._sample(input_ids, attention_mask, encoder_hidden_states, max_new_tokens, int(top_k), eos_token_id,
                                pad_token_id, sample_noise)
        dev = input_ids.device
        B, cur = input_ids.shape
        nb = int(num_beams)
        max_length = cur + int(max_new_tokens)
        ids = input_ids.repeat_interleave(nb, dim=0)
        mask = attention_mask.repeat_interleave(nb, dim=0)
        enc = encoder_hidden_states.repeat_interleave(nb, dim=0).contiguous() if encoder_hidden_states is not None else None
        beam_scores = torch.zeros(B, nb, dtype=torch.float32, device=dev)
        beam_scores[:, 1:] = -1e9
        beam_scores = beam_scores.view(-1)
        hyps = [_BeamHypotheses(nb, length_penalty) for _ in range(B)]
        done = [False] * B
        # every beam of a sample attends to the same condition tokens and beams are only ever reordered within their sample, so
        # the per-layer cross-attention K/V of `enc` are step-invariant: projected at the first step, reused afterwards
        kv_cache = {} if enc is not None else None
        while True:
            logits = self.next_token_logits(ids, mask, enc, kv_cache).float()
            scores = torch.log_softmax(logits, dim=-1) + beam_scores[:, None]
            V = scores.shape[-1]
            top_s, top_i = torch.topk(scores.view(B, nb * V), 2 * nb, dim=1, largest=True, sorted=True)
            top_s, top_i = top_s.cpu(), top_i.cpu()
            src_beam, tok = top_i // V, top_i % V
            ids_cpu = ids.cpu()
            cur_len = ids.shape[1] + 1
            nxt_s = torch.zeros(B, nb)
            nxt_t = torch.zeros(B, nb, dtype=torch.long)
            nxt_b = torch.zeros(B, nb, dtype=torch.long)
            for b in range(B):
                if done[b]:
                    nxt_t[b] = pad_token_id
                    continue
                k = 0
                for rank in range(2 * nb):
                    row = b * nb + int(src_beam[b, rank])
                    if eos_token_id is not None and int(tok[b, rank]) == eos_token_id:
                        if rank >= nb:
                            continue
                        hyps[b].add(ids_cpu[row].clone(), float(top_s[b, rank]))
                    else:
                        nxt_s[b, k], nxt_t[b, k], nxt_b[b, k] = top_s[b, rank], tok[b, rank], row
                        k += 1
                    if k == nb:
                        break
                done[b] = done[b] or hyps[b].is_done(float(top_s[b].max()), cur_len)
            beam_scores = nxt_s.view(-1).to(dev)
            ids = torch.cat([ids[nxt_b.view(-1).to(dev)], nxt_t.view(-1, 1).to(dev)], dim=1)
            mask = self.update_attention_mask(mask)
            if all(done) or ids.shape[1] >= max_length:
                break
        ids_cpu, fin = ids.cpu(), beam_scores.cpu()
        best = []
        for b in range(B):
            if not done[b]:
                for k in range(nb):
                    hyps[b].add(ids_cpu[b * nb + k], float(fin[b * nb + k]))
            best.append(max(hyps[b].beams, key=lambda h: h[0])[1])
        lens = [int(h.shape[0]) for h in best]
        width = min(max(lens) + 1, max_length)
        out = torch.full((B, width), pad_token_id if pad_token_id is not None else 0, dtype=torch.long)
        for b, h in enumerate(best):
            out[b, :lens[b]] = h
            if lens[b] < width:
                out[b, lens[b]] = eos_token_id
        return out.to(dev)


    def _sample(self, input_ids, attention_mask, enc, max_new_tokens, top_k, eos_token_id, pad_token_id, noise):
        """Top-k sampling as the reference's captioner_mode asks transformers 4.31 for it (vast.py:526-536: do_sample=True, top_k=10,
        temperature 1): per step the top_k logits are kept (TopKLogitsWarper), softmax over them, ONE draw per row; rows that have
        produced eos emit pad from then on; stop when every row has finished or max_length is reached.  The draw is inverse-CDF over
        the kept candidates in descending-score order with one uniform number per (row, step): `noise` [rows, max_new_tokens] injects
        them (parity tests), otherwise they come from torch's generator - the same distribution as torch.multinomial, not the same
        stream.  Device: model step + top-k; host: k candidates per row."""
        dev = input_ids.device
        B, cur = input_ids.shape
        max_length = cur + int(max_new_tokens)
        ids, mask = input_ids, attention_mask
        unfinished = torch.ones(B, dtype=torch.bool)
        kv_cache = {} if enc is not None else None
        if enc is not None:
            enc = enc.contiguous()
        step = 0
        while True:
            logits = self.next_token_logits(ids, mask, enc, kv_cache).float()
            top_s, top_i = torch.topk(logits, min(top_k, logits.shape[-1]), dim=-1, largest=True, sorted=True)
            probs = torch.softmax(top_s, dim=-1).cpu().double()
            top_i = top_i.cpu()
            u = noise[:, step].double().cpu() if noise is not None else torch.rand(B, dtype=torch.float64)
            cdf = probs.cumsum(-1)
            pick = (cdf < (u * cdf[:, -1])[:, None]).sum(-1).clamp_max(probs.shape[-1] - 1)
            tok = top_i[torch.arange(B), pick]
            if eos_token_id is not None:
                tok = torch.where(unfinished, tok, torch.full_like(tok, pad_token_id if pad_token_id is not None else 0))
                unfinished = unfinished & (tok != eos_token_id)
            ids = torch.cat([ids, tok.view(-1, 1).to(dev)], dim=1)
            mask = self.update_attention_mask(mask)
            step += 1
            if not bool(unfinished.any()) or ids.shape[1] >= max_length:
                break
        return ids


class _BeamHypotheses:
    """n-best list of finished hypotheses of one sample (transformers==4.31 BeamHypotheses, early_stopping=False)."""

    def __init__(self, num_beams, length_penalty):
        self.num_beams, self.length_penalty = num_beams, length_penalty
        self.beams, self.worst_score = [], 1e9

    def add(self, hyp, sum_logprobs):
        score = sum_logprobs / (hyp.shape[-1] ** self.length_penalty)
        if len(self.beams) < self.num_beams or score > self.worst_score:
            self.beams.append((score, hyp))
            if len(self.beams) > self.num_beams:
                order = sorted((s, i) for i, (s, _) in enumerate(self.beams))
                del self.beams[order[0][1]]
                self.worst_score = order[1][0]
            else:
                self.worst_score = min(score, self.worst_score)

    def is_done(self, best_sum_logprobs, cur_len):
        if len(self.beams) < self.num_beams:
            return False
        return self.worst_score >= best_sum_logprobs / cur_len ** self.length_penalty


def build_tokenizer():
    """BertTokenizer over the bert-base-uncased WordPiece vocabulary, with the special ids the reference sets
    (mico.py:109-113): bos=[CLS] 101, eos=[SEP] 102, pad=[PAD] 0, mask=[MASK] 103."""
    from transformers import BertTokenizer
    tok = BertTokenizer.from_pretrained(TOKENIZER_DIR)   # vocab.txt + tokenizer_config.json (mico.py:109)
    tok.bos_token_id = tok.convert_tokens_to_ids(["[CLS]"])[0]
    tok.eos_token_id = tok.convert_tokens_to_ids(["[SEP]"])[0]
    tok.pad_token_id = tok.convert_tokens_to_ids(["[PAD]"])[0]
    tok.mask_token_id = tok.convert_tokens_to_ids(["[MASK]"])[0]
    return tok
