"""MiCo.forward(batch, task, compute_loss) - the omni-modal alignment step.

Specification: the VAST sibling's trainer-facing forward (data/model/vast.py:317-348), forward_ret (:383-464: ITC with
label smoothing 0.1 against the all-gathered global batch, ITM with in-batch hard negatives) and forward_cap (:485-512:
causal masked-caption LM), generalised with MiCo's depth heads (model/mico.py:390,392,402,406).  These functions become
methods of mico_amd.model.mico.MiCo.

batch keys: vision_pixels [b,n,3,h,w] | audio_spectrograms [b,n,h,w] | depth_pixels [b,n,3,h,w] (any subset);
            raw_captions (list[str]) or input_ids/attention_mask [b,S].
            optional `_injected`: {subtask: {neg_cond_idx, neg_text_idx}, "cap": {masked_ids, labels}} replaces the RNG draws
            (torch.multinomial / TokenMasker) for parity tests; optional `_world`: simulated gathered tensors.
"""
import torch

from .. import distributed as D
from .. import functional as Fn, ops
from .. import runtime

COND_MODALITY = {"v": "vision", "a": "audio", "d": "depth"}
FUSED_HEADS = {"v": "contra_head_v", "a": "contra_head_a", "d": "contra_head_d", "s": "contra_head_s", "va": "contra_head_va",
               "vd": "contra_head_id", "vs": "contra_head_vs", "vas": "contra_head_vas"}
SUBTASKS = ("tv", "ta", "td", "ts", "tva", "tvd", "tvs", "tvas")    # vast.py's tv / ta / tva / tvs / tvas + MiCo's depth heads


def _tokens(self, batch):
    if "input_ids" in batch:
        return batch["input_ids"], batch["attention_mask"]
    if "caption_tokens" in batch:
        ct = batch["caption_tokens"]
        return ct.input_ids, ct.attention_mask
    dev = self.contra_temp.device
    tok = self.multimodal_encoder.tokenizer(batch["raw_captions"], padding="max_length", truncation=True,
                                            max_length=self.max_caption_len, return_tensors="pt")
    batch["input_ids"], batch["attention_mask"] = tok.input_ids.to(dev), tok.attention_mask.to(dev)
    return batch["input_ids"], batch["attention_mask"]


def encode_batch(self, batch):
    """batch_get of vast.py:81-314 for every modality present: ONE tower pass over all frames of all modalities (they share
    the ViT), pooled features, packed condition tensors, text feature."""
    enc = {}
    groups, meta = [], []
    for m, key in (("v", "vision_pixels"), ("a", "audio_spectrograms"), ("d", "depth_pixels")):
        if key not in batch:
            continue
        x = batch[key]
        b, n = x.shape[:2]
        g = x.reshape(b * n, 1, *x.shape[-2:]) if m == "a" else x.reshape(b * n, *x.shape[2:])
        groups.append(g)
        meta.append((m, b, n))
    if groups:
        # injected stochastic-depth multipliers ({modality: [depth, 2, b*n]}; parity tests) are laid out like the frames
        dps = (batch.get("_injected") or {}).get("drop_path_scale")
        if dps is not None:
            dps = torch.cat([dps[m].float().cpu() for m, _, _ in meta], dim=-1)
        if self.config.vision_encoder_type.startswith("swin"):
            # one tower pass as well; spectrograms take the reference's route of three identical channels (mico.py:139-140)
            frames = torch.cat([g.expand(-1, 3, -1, -1) if g.shape[1] == 1 else g for g in groups], dim=0)
            tokens = self.vision_encoder.forward_features(frames, drop_path_scale=dps)
        else:
            tokens = self.vision_encoder.visual.forward_groups(groups, drop_path_scale=dps)
        f0 = 0
        for m, b, n in meta:
            o = tokens[f0:f0 + b * n].view(b, n, *tokens.shape[-2:])
            f0 += b * n
            enc["output_" + m] = o
            enc["pooled_" + m] = self.pool_vision_for_contra(o)
            enc["condition_feats_" + m] = self._pack(COND_MODALITY[m], o)
    if "subtitle_ids" in batch or "raw_subtitles" in batch:   # vast.py:96-104,168-174: subtitles through the text BERT
        if "subtitle_ids" not in batch:
            dev = self.contra_temp.device
            tok = self.multimodal_encoder.tokenizer(batch["raw_subtitles"], padding="max_length", truncation=True,
                                                    max_length=self.max_subtitle_len, return_tensors="pt")
            batch["subtitle_ids"], batch["subtitle_mask"] = tok.input_ids.to(dev), tok.attention_mask.to(dev)
        sub = self.multimodal_encoder.bert(input_ids=batch["subtitle_ids"], attention_mask=batch["subtitle_mask"]).last_hidden_state
        enc["output_s"] = sub
        enc["pooled_s"] = self.pool_text_for_contra(sub)
        enc["condition_feats_s"] = self.get_multimodal_forward_input_subtitle(sub)
    if "input_ids" in batch or "raw_captions" in batch or "caption_tokens" in batch:
        ids, am = _tokens(self, batch)
        seq = self.multimodal_encoder.bert(input_ids=ids, attention_mask=am).last_hidden_state
        enc["caption_output"] = seq
        enc["feat_t"] = Fn.l2_normalize(self.contra_head_t(self.pool_text_for_contra(seq)))
    return enc


def _feat_cond(self, enc, cond):
    pooled = torch.cat([enc["pooled_" + m] for m in cond], dim=1) if len(cond) > 1 else enc["pooled_" + cond]
    return Fn.l2_normalize(getattr(self, FUSED_HEADS[cond])(pooled))


def _condition_feats(self, enc, cond):
    if len(cond) == 1:
        return enc["condition_feats_" + cond]
    return torch.cat([enc["condition_feats_" + m] for m in cond], dim=1)


def _forward_ret(self, batch, enc, subtasks, itm=True, deferred=None):
    """ITC + ITM of vast.py:395-457.  itm=False: the contrastive objective alone (step-A of SURVEY.md section 8d, BASELINE configs[1]) - the
    ITM hard-negative passes are not run and no "loss_itm" is returned (task prefix "itc%...", an addition of this repo).
    deferred (dict, staged differentiation - forward(backward_scale=...)): the ITM passes are not run here either; their inputs - the hard-negative
    draws (same random numbers in the same order as the direct form), the triplet's token ids and masks - are left in deferred[subtask] for
    _staged_groups, and only "loss_itc" is returned."""
    inj = batch.get("_injected", {})
    world = batch.get("_world")
    ids, am = _tokens(self, batch)
    rank = world["rank"] if world else D.rank()
    bs = ids.shape[0]
    feat_t = enc["feat_t"]
    feats = {st: _feat_cond(self, enc, st[1:]) for st in subtasks}
    if world:
        feat_t_all, ids_all, mask_all = world["feat_t_all"], world["ids_all"], world["mask_all"]
        feats_all = {st: world[f"feat_{st[1:]}_all"] for st in subtasks}
    else:   # ONE packed collective for every small per-step tensor (reference: 3 + len(subtasks) all_gathers)
        packed = D.packed_all_gather([feat_t, ids, am] + [feats[st] for st in subtasks])
        feat_t_all, ids_all, mask_all = packed[0], packed[1], packed[2]
        feats_all = dict(zip(subtasks, packed[3:]))
    targets = torch.arange(rank * bs, rank * bs + bs, device=ids.device)
    loss_itc, loss_itm = [], []
    for st in subtasks:
        fc, fc_all = feats[st], feats_all[st]
        sim_c2t = Fn.matmul_nt(fc, feat_t_all) / self.contra_temp                       # vast.py:405-408
        sim_t2c = Fn.matmul_nt(feat_t, fc_all) / self.contra_temp
        loss_itc.append((Fn.cross_entropy(sim_c2t, targets, 0.1) + Fn.cross_entropy(sim_t2c, targets, 0.1)) / 2)
        if not itm:
            continue
        # ---- ITM hard negatives (vast.py:421-457) ----
        cond = _condition_feats(self, enc, st[1:]) if deferred is None else None
        if st in inj:
            neg_c, neg_t = inj[st]["neg_cond_idx"].to(ids.device), inj[st]["neg_text_idx"].to(ids.device)
        else:
            # one kernel per direction: softmax + 1e-4, own-rank diagonal zeroed, inverse-CDF draw per row (mico_itm_sample) - the
            # reference loops over rows with a .item() host sync each (vast.py:428-440); `_itm_uniform` injects the random numbers
            un = inj.get("_itm_uniform", {}).get(st)
            u_c, u_t = (un[0].to(ids.device), un[1].to(ids.device)) if un is not None else (
                torch.rand(bs, device=ids.device), torch.rand(bs, device=ids.device))
            neg_c = ops.itm_sample(sim_t2c.detach(), rank * bs, u_c.float())
            neg_t = ops.itm_sample(sim_c2t.detach(), rank * bs, u_t.float())
        ids1 = torch.cat((ids, ids, ids_all[neg_t]), dim=0)
        am1 = torch.cat((am, am, mask_all[neg_t]), dim=0)
        if deferred is not None:
            deferred[st] = dict(neg_c=neg_c, ids1=ids1, am1=am1, fetch=world[f"cond_{st[1:]}_fetch"] if world else D.fetch_rows)
            continue
        if world:
            cond_neg = world[f"cond_{st[1:]}_fetch"](cond, neg_c)
        else:
            cond_neg = D.fetch_rows(cond, neg_c)
        if _share_cross_kv(self):
            # the triplet [own | hard negative | own] holds the batch's own condition tokens twice and the captioning pass reads
            # them again: their K/V projections are computed once per step (functional.CrossKVFn) and kept for _forward_cap
            kv = self.multimodal_encoder.bert.project_cross_kv(cond, cond_neg)
            enc.setdefault("_cross_kv", {})[st[1:]] = kv
            out = self.multimodal_encoder.bert(input_ids=ids1, attention_mask=am1, cross_kv=kv).last_hidden_state
        else:
            cond3 = torch.cat((cond, cond_neg, cond), dim=0)
            out = self.multimodal_encoder.bert(input_ids=ids1, attention_mask=am1, encoder_hidden_states=cond3).last_hidden_state
        logits = self.itm_head(out[:, 0])
        gt = torch.zeros(bs * 3, dtype=torch.long, device=ids.device)
        gt[:bs] = 1
        loss_itm.append(self.itm_ratio * Fn.cross_entropy(logits, gt))
    if not itm or deferred is not None:
        return {"loss_itc": sum(loss_itc) / len(loss_itc)}
    return {"loss_itc": sum(loss_itc) / len(loss_itc), "loss_itm": sum(loss_itm) / len(loss_itm)}


def _itm_loss(self, kv, ids1, am1):
    """The ITM pass over the triplet [own | hard negative | own] against a shared K/V memory, and its loss (vast.py:438-457)."""
    bs = ids1.shape[0] // 3
    out = self.multimodal_encoder.bert(input_ids=ids1, attention_mask=am1, cross_kv=kv).last_hidden_state
    logits = self.itm_head(out[:, 0])
    gt = torch.zeros(bs * 3, dtype=torch.long, device=ids1.device)
    gt[:bs] = 1
    return self.itm_ratio * Fn.cross_entropy(logits, gt)


def _share_cross_kv(self):
    """Share the cross-attention K/V projections between the passes of a training step (runtime.CFG.share_cross_kv)."""
    return runtime.CFG.share_cross_kv and torch.is_grad_enabled()


def _cap_inputs(self, batch):
    """masked token ids, labels (TokenMasker at 0.6 or the injected draw) and the causal 3-D mask of the captioning pass (vast.py:489-499)."""
    inj = batch.get("_injected", {})
    ids, am = _tokens(self, batch)
    if "cap" in inj:
        masked_ids, labels = inj["cap"]["masked_ids"].to(ids.device), inj["cap"]["labels"].to(ids.device)
    else:
        masked_ids, labels = self.text_masker(ids, 0.6)
    S = am.shape[1]
    m3 = torch.tril(am.unsqueeze(1).expand(-1, S, -1)).contiguous()                       # vast.py:497-499
    return masked_ids, labels, m3


def _forward_cap(self, batch, enc, subtasks):
    masked_ids, labels, m3 = _cap_inputs(self, batch)
    losses = []
    for st in subtasks:
        kv = enc.get("_cross_kv", {}).get(st[1:]) if _share_cross_kv(self) else None
        if kv is not None:   # the retrieval branch of this step already projected these condition tokens
            losses.append(self.multimodal_encoder(input_ids=masked_ids, attention_mask=m3, cross_kv=(kv[0], None), labels=labels).loss)
            continue
        cond = _condition_feats(self, enc, st[1:])
        losses.append(self.multimodal_encoder(input_ids=masked_ids, attention_mask=m3, encoder_hidden_states=cond,
                                              labels=labels).loss)
    return {"loss_cap": sum(losses) / len(losses)}


class _StagedLoss(torch.autograd.Function):
    """The hand-over of a staged step (forward(backward_scale=...)): `value` is a loss that has ALREADY been differentiated through BERT inside
    forward, `grads[i]` = d(scale * sum of the staged losses) / d(tensors[i]) for the tower-side tensors the BERT passes read (the per-modality
    condition tokens).  The caller's backward of scale * (sum of the returned losses) arrives here with grad_output = scale and hands those
    gradients to the towers' graph (multiplied by grad_output / scale on the device - 1 when the caller keeps its side of the contract)."""

    @staticmethod
    def forward(ctx, value, scale, n, *tg):
        ctx.scale, ctx.n, ctx.grads = scale, n, tg[n:]
        return value.detach().clone()

    @staticmethod
    def backward(ctx, go):
        if ctx.grads is None:
            raise RuntimeError("_StagedLoss: a second backward through a staged step - its BERT side was differentiated inside the forward and the "
                               "condition-token gradients were handed over (scaled in place) by the first one; run the forward again")
        f = (go / ctx.scale).to(torch.float32)
        grads, ctx.grads = ctx.grads, None
        return (None, None, None) + tuple(g.mul_(f) for g in grads) + (None,) * ctx.n


def _forward_staged(self, batch, task, enc, scale):
    """forward(compute_loss=True) with the BERT passes differentiated ONE CONDITION SET AT A TIME inside the forward (round 6).
    The direct form builds every ITM / captioning graph, their cross-attention K/V memories and, in the backward, their gradients on top of the
    towers' complete activation stash (profiles/r05_mem_trace.txt: the step's peak is the second triplet's BertFn.backward).  Here the towers'
    condition tokens are cut out of the graph (detached leaves, one per modality), and for each condition set (e.g. "va": the tva triplet + the
    captioning pass that shares its K/V memory) the losses are built AND differentiated at once with the factor the caller will apply (`scale`:
    GradScaler's loss scale, 1.0 without one) - BERT-side parameter gradients go to .grad right away, the token gradients accumulate on the
    leaves, and the set's graph, K/V memory and gradient buffers are gone before the next set is built.  The returned losses carry one
    _StagedLoss node that hands the accumulated token gradients to the towers when the caller differentiates the sum.  Same loss values and
    gradients as the direct form (same kernels on the same numbers; only the order in which autograd sums the token gradients differs);
    random draws (hard negatives, token masks) happen in the direct form's order, BERT's per-pass dropout seeds are drawn in pass order, which
    differs.  Contract: the caller calls backward() ONCE on scale * (unit-weight sum of the returned losses), after zero_grad - parameter
    gradients of the BERT side are already in .grad when forward returns."""
    ret_sub, cap_sub, out = [], [], {}
    deferred = {}
    for t in task.split("_"):
        subtasks = t.split("%")[1:]
        for st in subtasks:
            assert st in SUBTASKS, st
        if t.startswith("ret"):
            out.update(_forward_ret(self, batch, enc, subtasks, deferred=deferred))
            ret_sub += subtasks
        elif t.startswith("itc"):
            out.update(_forward_ret(self, batch, enc, subtasks, itm=False))
        elif t.startswith("cap"):
            cap_sub += subtasks
        else:
            raise NotImplementedError(t)
    cap_in = _cap_inputs(self, batch) if cap_sub else None
    keys = []
    for st in ret_sub + cap_sub:
        if st[1:] not in keys:
            keys.append(st[1:])
    leaves = {}
    sums = {"loss_itm": None, "loss_cap": None}
    from ..distributed import staged_backward
    for key in keys:
        for m in key:
            if m not in leaves:
                leaves[m] = enc["condition_feats_" + m].detach().requires_grad_(True)
        cond = torch.cat([leaves[m] for m in key], dim=1) if len(key) > 1 else leaves[key]
        total, kv = None, None
        itm_sts = [st for st in ret_sub if st[1:] == key]
        for st in itm_sts:
            d = deferred[st]
            kv = self.multimodal_encoder.bert.project_cross_kv(cond, d["fetch"](cond, d["neg_c"]))
            l = _itm_loss(self, kv, d["ids1"], d["am1"]) / len(ret_sub)
            sums["loss_itm"] = l.detach() if sums["loss_itm"] is None else sums["loss_itm"] + l.detach()
            total = l if total is None else total + l
        for st in cap_sub:
            if st[1:] != key:
                continue
            masked_ids, labels, m3 = cap_in
            if kv is not None and len(itm_sts) == 1:     # the retrieval branch of this set projected these condition tokens
                l = self.multimodal_encoder(input_ids=masked_ids, attention_mask=m3, cross_kv=(kv[0], None), labels=labels).loss
            else:
                l = self.multimodal_encoder(input_ids=masked_ids, attention_mask=m3, encoder_hidden_states=cond, labels=labels).loss
            l = l / len(cap_sub)
            sums["loss_cap"] = l.detach() if sums["loss_cap"] is None else sums["loss_cap"] + l.detach()
            total = l if total is None else total + l
        del kv
        runtime.mem_trace("staged set " + key + ": graph built")
        with staged_backward():
            torch.autograd.backward(total * scale)
        del total, l, cond
        runtime.mem_trace("staged set " + key)
    first = True
    for k, v in sums.items():
        if v is None:
            continue
        if first and leaves:      # ONE node carries every condition-token gradient
            ms = list(leaves)
            v = _StagedLoss.apply(v, float(scale), len(ms), *[enc["condition_feats_" + m] for m in ms], *[leaves[m].grad for m in ms])
            first = False
        out[k] = v
    return out


def forward(self, batch, task, compute_loss=True, backward_scale=None):
    """Returns {"loss_itc", "loss_itm", "loss_cap"} for task strings like "ret%tva%tv_cap%tva" (vast.py:317-348).
    backward_scale (float; None = the direct form): staged differentiation, see _forward_staged - the factor the caller multiplies the summed
    losses with before its single backward() (1.0, or GradScaler.get_scale())."""
    batch = dict(batch) if not isinstance(batch, dict) else batch
    runtime.mem_trace("step start")
    kinds = [t.split("%")[0] for t in task.split("_")]
    # (a task string that names a branch twice keeps the direct form, whose later branch overwrites the earlier one's losses - vast.py:317-348)
    staged = backward_scale is not None and compute_loss and torch.is_grad_enabled() and _share_cross_kv(self) and len(set(kinds)) == len(kinds)
    runtime.step_staged = staged      # (functional.tower_plan: a staged step needs less memory next to the towers' saved activations)
    try:
        enc = encode_batch(self, batch)
    finally:
        runtime.step_staged = False   # (the plan is made inside the tower's forward: a tower pass outside MiCo.forward is priced as a direct step)
    runtime.mem_trace("after encode_batch")
    if staged:
        return _forward_staged(self, batch, task, enc, float(backward_scale))
    out = {}
    for t in task.split("_"):
        subtasks = t.split("%")[1:]
        for st in subtasks:
            assert st in SUBTASKS, st
        if t.startswith("ret"):
            if compute_loss:
                out.update(_forward_ret(self, batch, enc, subtasks))
                runtime.mem_trace("after forward_ret")
            else:   # evaluation dict of vast.py:466-483
                ids, am = _tokens(self, batch)
                out.update(feat_t=enc["feat_t"], input_ids=ids, attention_mask=am)
                for st in subtasks:
                    out[f"feat_cond_{st}"] = _feat_cond(self, enc, st[1:])
                    out[f"condition_feats_{st}"] = _condition_feats(self, enc, st[1:])
        elif t.startswith("itc"):      # contrastive objective only (no ITM passes): step-A of SURVEY.md section 8d
            assert compute_loss, "itc%... is a training objective"
            out.update(_forward_ret(self, batch, enc, subtasks, itm=False))
        elif t.startswith("cap"):
            if compute_loss:
                out.update(_forward_cap(self, batch, enc, subtasks))
                runtime.mem_trace("after forward_cap")
            else:   # evaluation dict of vast.py:513-547: beam-search captions per sub-task (captioner_mode sampling is not provided)
                tk = self.multimodal_encoder.tokenizer
                for st in subtasks:
                    cond = _condition_feats(self, enc, st[1:])
                    if self.config.get("captioner_mode", False):
                        # vast.py:519-536: generate_nums sampled captions per sample (top-k 10 sampling), rows sample-major
                        gn = int(self.config.generate_nums)
                        cond = cond.unsqueeze(1).expand(-1, gn, -1, -1).reshape(-1, *cond.shape[1:]).contiguous()
                        init = torch.full((cond.shape[0], 1), tk.bos_token_id, dtype=torch.long, device=cond.device)
                        ids = self.multimodal_encoder.generate(input_ids=init, attention_mask=init.new_ones(cond.shape[0], 1, 1),
                                                               do_sample=True, top_k=10, encoder_hidden_states=cond,
                                                               max_new_tokens=self.max_caption_len, eos_token_id=tk.sep_token_id,
                                                               pad_token_id=tk.pad_token_id,
                                                               sample_noise=(batch.get("_injected") or {}).get("sample_noise"))
                        out[f"generated_captions_{st}"] = tk.batch_decode(ids[:, 1:], skip_special_tokens=True)
                        continue
                    init = torch.full((cond.shape[0], 1), tk.bos_token_id, dtype=torch.long, device=cond.device)
                    ids = self.multimodal_encoder.generate(input_ids=init, attention_mask=init.new_ones(cond.shape[0], 1, 1),
                                                           encoder_hidden_states=cond, max_new_tokens=self.max_caption_len,
                                                           num_beams=self.beam_size, eos_token_id=tk.sep_token_id,
                                                           pad_token_id=tk.pad_token_id, length_penalty=0.6)
                    out[f"generated_captions_{st}"] = tk.batch_decode(ids[:, 1:], skip_special_tokens=True)
        else:
            raise NotImplementedError(t)
    return out
