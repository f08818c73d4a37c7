"""TEST INFRASTRUCTURE - generates tests/golden/*.pt by running the REFERENCE (imported from /root/reference through
oracle/ref_import.py) on CPU fp32.  Build container only; the fixtures it writes are plain tensors (inputs are
regenerated from seeds by mico_amd.weights, only expected outputs are stored).

    python -m oracle.make_golden            # all fixtures
    python -m oracle.make_golden vit bert   # a subset

Reference call sites exercised: model/mico.py:115-248 (encoders, pooling, condition packing, heads),
model/evaclip/eva_vit_model.py:611-659, model/bert.py:785-916,1047-1097; the loss composition follows
data/model/vast.py:383-464 (forward_ret) and :485-512 (forward_cap) line by line with the reference's modules, with the
torch.multinomial / TokenMasker draws replaced by recorded indices.
"""
import os
import random
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_import  # noqa: E402
from mico_amd.weights import synth_state_dict, synth_inputs  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def fill(model, seed=0):
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synth_state_dict(shapes, seed)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    # transformers==4.31 ties the LM decoder to the word embeddings (model/bert.py:1038-1041); 5.x does not -> tie here
    if hasattr(model, "multimodal_encoder"):
        me = model.multimodal_encoder
        me.cls.predictions.decoder.weight = me.bert.embeddings.word_embeddings.weight
        me.cls.predictions.decoder.bias = me.cls.predictions.bias
    return sd


def grad_digest(t):
    f = t.detach().flatten().float()
    return dict(head=f[:256].clone(), norm=f.norm().clone(), asum=f.abs().sum().clone(), shape=tuple(t.shape))


def vit_fixture(vtype, depth, tag):
    torch.manual_seed(0)
    m = ref_import.build_mico(vtype, depth=depth)
    fill(m)
    vis = m.vision_encoder.visual
    g = torch.Generator().manual_seed(77)
    x = torch.randn((2, 3, 224, 224), generator=g)
    taps = []
    hooks = [blk.register_forward_hook(lambda mod, i, o: taps.append(o.detach())) for blk in vis.blocks]
    for p in m.parameters():
        p.requires_grad_(True)
    out = vis(x, return_all_features=True)
    for h in hooks:
        h.remove()
    w = torch.randn(out.shape, generator=g) / out.numel() ** 0.5
    (out * w).sum().backward()
    names = ["patch_embed.proj.weight", "patch_embed.proj.bias", "cls_token", "pos_embed", "norm.weight", "norm.bias"]
    for i in sorted({0, depth - 1}):
        for k, _ in vis.blocks[i].named_parameters():
            names.append(f"blocks.{i}.{k}")
    named = dict(vis.named_parameters())
    fx = dict(
        out=out.detach().clone(),
        tap_mean=torch.stack([t.mean() for t in taps]), tap_amax=torch.stack([t.abs().max() for t in taps]),
        tap_rows=torch.stack([t[:, [0, 1, 100]] for t in taps]),
        grads={n: grad_digest(named[n].grad) for n in names if named[n].grad is not None},
        meta=dict(vtype=vtype, depth=depth, input_seed=77),
    )
    torch.save(fx, os.path.join(OUT, f"vit_{tag}.pt"))
    print("wrote", f"vit_{tag}.pt", tuple(out.shape), float(out.abs().max()))
    return m


def vit_bige_fixture(depth=2):
    """EVA02-CLIP-bigE-14-plus (mico.py:341-344), the post-norm tower: the reference's own EVAVisionTransformer with the arguments
    model/evaclip/model.py:105-131 derives from model_configs/EVA02-CLIP-bigE-14-plus.json, truncated to `depth` blocks at construction (the
    full MiCo wrapper would allocate 64 blocks = 4.3 B parameters first); same inputs / weights / digests as vit_fixture."""
    from functools import partial
    ns = ref_import.load()
    torch.manual_seed(0)
    vis = ns.ref_eva.EVAVisionTransformer(
        img_size=224, patch_size=14, num_classes=1024, use_mean_pooling=False, init_values=None, patch_dropout=0.0, embed_dim=1792,
        depth=depth, num_heads=1792 // 112, mlp_ratio=8.571428571428571, qkv_bias=True, drop_path_rate=0.0,
        norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), xattn=True, rope=False, postnorm=True, pt_hw_seq_len=16, intp_freq=False,
        naiveswiglu=False, subln=False).eval()
    shapes = {"vision_encoder.visual." + k: tuple(v.shape) for k, v in vis.state_dict().items()}
    sd = synth_state_dict(shapes, 0)
    vis.load_state_dict({k[len("vision_encoder.visual."):]: v for k, v in sd.items()}, strict=True)
    g = torch.Generator().manual_seed(77)
    x = torch.randn((2, 3, 224, 224), generator=g)
    taps = []
    hooks = [blk.register_forward_hook(lambda mod, i, o: taps.append(o.detach())) for blk in vis.blocks]
    for p in vis.parameters():
        p.requires_grad_(True)
    out = vis(x, return_all_features=True)
    for h in hooks:
        h.remove()
    w = torch.randn(out.shape, generator=g) / out.numel() ** 0.5
    (out * w).sum().backward()
    named = dict(vis.named_parameters())
    fx = dict(
        out=out.detach().clone(),
        tap_mean=torch.stack([t.mean() for t in taps]), tap_amax=torch.stack([t.abs().max() for t in taps]),
        tap_rows=torch.stack([t[:, [0, 1, 100]] for t in taps]),
        grads={n: grad_digest(p.grad) for n, p in named.items() if p.grad is not None and not n.startswith("head.")},
        meta=dict(vtype="evaclip02_bige", depth=depth, input_seed=77),
    )
    torch.save(fx, os.path.join(OUT, "vit_bige_d2.pt"))
    print("wrote vit_bige_d2.pt", tuple(out.shape), float(out.abs().max()), len(fx["grads"]), "gradient digests")


def bert_fixture(m):
    me = m.multimodal_encoder
    for p in m.parameters():
        p.grad = None
    g = torch.Generator().manual_seed(5)
    b, S, E = 3, 16, 40
    ids = torch.randint(1000, 30000, (b, S), generator=g)
    ids[:, 0] = 101
    lens = torch.tensor([16, 9, 5])
    mask = (torch.arange(S)[None] < lens[:, None]).long()
    ids = ids * mask
    cond = torch.randn((b, E, 768), generator=g)
    fx = dict(meta=dict(b=b, S=S, E=E, seed=5, lens=lens))
    o = me(input_ids=ids, attention_mask=mask)
    def top2(tag, logits):
        """argmax ids plus how decided each one is: gap between the two largest logits of the position relative to the largest
        |logit| of the tensor - the parity test demands bit-exact ids wherever the gap exceeds the 16-bit logit resolution."""
        t, i = logits.detach().float().topk(2, dim=-1)
        fx[tag + "_argmax"] = logits.argmax(-1).clone()
        fx[tag + "_top2_ids"] = i.clone()
        fx[tag + "_top2_gap"] = ((t[..., 0] - t[..., 1]) / logits.detach().abs().max()).clone()

    fx["self_seq"] = o.sequence_output.detach().clone()
    top2("self", o.logits)
    o = me(input_ids=ids, attention_mask=mask, encoder_hidden_states=cond)
    fx["cross_seq"] = o.sequence_output.detach().clone()
    top2("cross", o.logits)
    m3 = torch.tril(mask.unsqueeze(1).expand(-1, S, -1).clone())
    labels = torch.full((b, S), -100)
    labels[0, 3], labels[0, 7], labels[1, 2], labels[2, 1] = 2000, 1037, 30521, 999
    cond_r = cond.clone().requires_grad_(True)
    o = me(input_ids=ids, attention_mask=m3, encoder_hidden_states=cond_r, labels=labels)
    fx["causal_seq"] = o.sequence_output.detach().clone()
    fx["causal_loss"] = o.loss.detach().clone()
    top2("causal", o.logits)
    fx["labels"] = labels
    o.loss.backward()
    named = dict(me.named_parameters())
    gn = ["bert.embeddings.word_embeddings.weight", "bert.embeddings.position_embeddings.weight",
          "bert.embeddings.LayerNorm.weight", "bert.encoder.layer.0.attention.self.query.weight",
          "bert.encoder.layer.0.crossattention.self.key.weight", "bert.encoder.layer.0.crossattention.self.value.bias",
          "bert.encoder.layer.11.output.dense.weight", "bert.encoder.layer.11.output.LayerNorm.bias",
          "cls.predictions.transform.dense.weight", "cls.predictions.bias"]
    fx["causal_grads"] = {n: grad_digest(named[n].grad) for n in gn}
    fx["causal_dcond"] = cond_r.grad.detach().clone()
    torch.save(fx, os.path.join(OUT, "bert.pt"))
    print("wrote bert.pt", float(fx["causal_loss"]))


def facade_fixture(m, tag, b=2):
    """inference_demo.py:128-158 style calls + every pool / condition-pack branch."""
    for p in m.parameters():
        p.grad = None
    fx = dict(meta=dict(b=b, tag=tag))
    cfgs = {"n1": dict(b=b, vision=1, audio=1, depth=1, S=20), "n4": dict(b=b, vision=4, audio=4, depth=1, S=20),
            "n3": dict(b=b, vision=3, audio=2, depth=1, S=20)}   # n3: frame embedding nearest-interp branch (8->3, 4->2)
    with torch.no_grad():
        for name, c in cfgs.items():
            inp = synth_inputs(c, seed=100)
            vo = m.forward_vision_encoder(inp["vision_pixels"])
            ao = m.forward_audio_encoder(inp["audio_spectrograms"])
            do = m.forward_depth_encoder(inp["depth_pixels"])
            r = {}
            r["feat_v"] = F.normalize(m.contra_head_v(m.pool_vision_for_contra(vo)), dim=-1)
            r["feat_a"] = F.normalize(m.contra_head_a(m.pool_audio_for_contra(ao)), dim=-1)
            r["feat_d"] = F.normalize(m.contra_head_d(m.pool_depth_for_contra(do)), dim=-1)
            r["feat_va"] = F.normalize(m.contra_head_va(torch.cat((m.pool_vision_for_contra(vo), m.pool_audio_for_contra(ao)), 1)), dim=-1)
            r["feat_vd"] = F.normalize(m.contra_head_id(torch.cat((m.pool_vision_for_contra(vo), m.pool_depth_for_contra(do)), 1)), dim=-1)
            to = m.forward_multimodal_encoder(inp["input_ids"], inp["attention_mask"]).sequence_output
            r["feat_t"] = F.normalize(m.contra_head_t(m.pool_text_for_contra(to)), dim=-1)
            r["sim_t2v"] = r["feat_t"] @ r["feat_v"].t()
            for pv in (False, True):
                m.config.pool_video = pv
                cv = m.get_multimodal_forward_input_vision(vo)
                ca = m.get_multimodal_forward_input_audio(ao)
                cd = m.get_multimodal_forward_input_depth(do)
                k = "pv" if pv else "full"
                r[f"cond_v_{k}_rows"] = cv[:, [0, 1, cv.shape[1] - 1]].clone()
                r[f"cond_v_{k}_sum"] = cv.sum((1, 2))
                r[f"cond_a_{k}_rows"] = ca[:, [0, 1, ca.shape[1] - 1]].clone()
                r[f"cond_d_{k}_rows"] = cd[:, [0, 1, cd.shape[1] - 1]].clone()
                out = m.forward_multimodal_encoder(inp["input_ids"], inp["attention_mask"], cv).sequence_output
                r[f"itm_score_{k}"] = F.softmax(m.itm_head(out[:, 0]), dim=1)[:, 1]
            m.config.pool_video = False
            r["vision_out_rows"] = vo[:, :, [0, 1, 50]].clone()
            r["audio_out_rows"] = ao[:, :, [0, 1, 50]].clone()
            fx[name] = r
    torch.save(fx, os.path.join(OUT, f"facade_{tag}.pt"))
    print("wrote", f"facade_{tag}.pt")


def loss_fixture(m, tag, b=4):
    """ITC + ITM + CAP for task 'ret%tva%tv_cap%tva' on a W=1 and a simulated W=2 world (two virtual ranks evaluated with
    the reference modules; rank r's loss uses gathered = cat(rank0, rank1) exactly as concat_all_gather would)."""
    itm_ratio = m.itm_ratio
    fx = dict(meta=dict(b=b, tag=tag, task="ret%tva%tv_cap%tva", itm_ratio=itm_ratio))
    rng = random.Random(11)
    from oracle.mico_oracle import token_masker

    def encode(inp):
        vo = m.forward_vision_encoder(inp["vision_pixels"])
        ao = m.forward_audio_encoder(inp["audio_spectrograms"])
        to = m.multimodal_encoder.bert(input_ids=inp["input_ids"], attention_mask=inp["attention_mask"]).last_hidden_state
        e = dict(vo=vo, ao=ao)
        e["feat_t"] = F.normalize(m.contra_head_t(m.pool_text_for_contra(to)), dim=-1)
        pv, pa = m.pool_vision_for_contra(vo), m.pool_audio_for_contra(ao)
        e["feat_v"] = F.normalize(m.contra_head_v(pv), dim=-1)
        e["feat_va"] = F.normalize(m.contra_head_va(torch.cat((pv, pa), 1)), dim=-1)
        cv, ca = m.get_multimodal_forward_input_vision(vo), m.get_multimodal_forward_input_audio(ao)
        e["cond_v"], e["cond_va"] = cv, torch.cat((cv, ca), dim=1)
        return e

    def rank_loss(inp, e, world, rank, inj):
        ids, am = inp["input_ids"], inp["attention_mask"]
        bs = ids.shape[0]
        l_itc, l_itm = [], []
        for st in ("tva", "tv"):
            c = st[1:]
            fc = e["feat_" + c]
            sim_c2t = fc @ world["feat_t_all"].t() / m.contra_temp                      # vast.py:405-408
            sim_t2c = e["feat_t"] @ world[f"feat_{c}_all"].t() / m.contra_temp
            targets = torch.arange(rank * bs, rank * bs + bs)
            l_itc.append((F.cross_entropy(sim_c2t, targets, label_smoothing=0.1)
                          + F.cross_entropy(sim_t2c, targets, label_smoothing=0.1)) / 2)   # :411-415
            cond, cond_all = e["cond_" + c], world[f"cond_{c}_all"]
            nci, nti = inj[st]["neg_cond_idx"], inj[st]["neg_text_idx"]
            ids1 = torch.cat((ids, ids, world["ids_all"][nti]), 0)                           # :445-448
            am1 = torch.cat((am, am, world["mask_all"][nti]), 0)
            cf = torch.cat((cond, cond_all[nci], cond), 0)
            out = m.multimodal_encoder.bert(input_ids=ids1, attention_mask=am1, encoder_hidden_states=cf).last_hidden_state
            logits = m.itm_head(out[:, 0])
            gt = torch.zeros(bs * 3, dtype=torch.long)
            gt[:bs] = 1
            l_itm.append(itm_ratio * F.cross_entropy(logits, gt))                            # :453-457
        S = am.shape[1]
        m3 = torch.tril(am.unsqueeze(1).expand(-1, S, -1).clone())                            # :497-499
        cap = m.multimodal_encoder(input_ids=inj["cap"]["masked_ids"], attention_mask=m3,
                                   encoder_hidden_states=e["cond_va"], labels=inj["cap"]["labels"]).loss
        return dict(loss_itc=sum(l_itc) / 2, loss_itm=sum(l_itm) / 2, loss_cap=cap)

    def make_inj(inp, e, world, rank):
        bs = inp["input_ids"].shape[0]
        inj = {}
        for st in ("tva", "tv"):
            c = st[1:]
            with torch.no_grad():
                w_t2c = F.softmax(e["feat_t"] @ world[f"feat_{c}_all"].t() / m.contra_temp, dim=1) + 1e-4   # :423-427
                w_t2c[:, rank * bs: rank * bs + bs].fill_diagonal_(0)
                w_c2t = F.softmax(e["feat_" + c] @ world["feat_t_all"].t() / m.contra_temp, dim=1) + 1e-4
                w_c2t[:, rank * bs: rank * bs + bs].fill_diagonal_(0)
            gg = torch.Generator().manual_seed(1000 + rank + len(st))
            inj[st] = dict(neg_cond_idx=torch.multinomial(w_t2c, 1, generator=gg).view(-1),
                           neg_text_idx=torch.multinomial(w_c2t, 1, generator=gg).view(-1),
                           w_t2c=w_t2c.clone(), w_c2t=w_c2t.clone())
        mi, lab = token_masker(inp["input_ids"], 0.6, rng)
        inj["cap"] = dict(masked_ids=mi, labels=lab)
        return inj

    gnames = ["contra_temp", "contra_head_t.linear.weight", "contra_head_va.weight", "itm_head.linear2.weight",
              "hidden_trans_vision_multimodal.0.weight", "vision_frame_embedding", "audio_type_embeddings",
              "vision_encoder.visual.patch_embed.proj.weight", "vision_encoder.visual.cls_token",
              "vision_encoder.visual.blocks.0.norm1.weight", "vision_encoder.visual.blocks.1.mlp.{}.weight",
              "multimodal_encoder.bert.embeddings.word_embeddings.weight",
              "multimodal_encoder.bert.encoder.layer.5.crossattention.self.key.weight",
              "multimodal_encoder.cls.predictions.bias"]
    named = dict(m.named_parameters())
    gnames = [n.format("w3" if "vision_encoder.visual.blocks.1.mlp.w3.weight" in named else "fc2") for n in gnames]

    for W in (1, 2, 4):
        inputs = [synth_inputs(dict(b=b, vision=2, audio=1, S=12), seed=1234 + r) for r in range(W)]
        encs = [encode(i) for i in inputs]
        world = dict(feat_t_all=torch.cat([e["feat_t"] for e in encs]).detach(),
                     ids_all=torch.cat([i["input_ids"] for i in inputs]), mask_all=torch.cat([i["attention_mask"] for i in inputs]))
        for c in ("v", "va"):
            world[f"feat_{c}_all"] = torch.cat([e["feat_" + c] for e in encs]).detach()
            # all_gather_with_grad: grads flow to every rank's condition_feats (distributed.py:12-47); for rank 0's
            # loss only rank 0's parameters' grads are recorded below, matching one process of the DDP job before the
            # gradient all-reduce.  Remote rows enter as constants here (their grad contribution belongs to the other
            # rank's backward and reaches the parameters through DDP's all-reduce).
            world[f"cond_{c}_all"] = torch.cat([encs[0]["cond_" + c]] + [e["cond_" + c].detach() for e in encs[1:]])
        inj = make_inj(inputs[0], encs[0], world, 0)
        for p in m.parameters():
            p.grad = None
        losses = rank_loss(inputs[0], encs[0], world, 0, inj)
        total = sum(losses.values())
        total.backward()
        r = dict(losses={k: v.detach().clone() for k, v in losses.items()}, inj=inj,
                 grads={n: grad_digest(named[n].grad) for n in gnames},
                 world={k: v.detach().clone() for k, v in world.items() if not k.startswith("cond_")},
                 feat_t=encs[0]["feat_t"].detach().clone(), feat_va=encs[0]["feat_va"].detach().clone())
        if W >= 2:   # the remote ranks' condition rows that may be fetched as negatives (recomputed from seeds in tests)
            r["remote_cond_va_sum"] = torch.cat([e["cond_va"].detach().sum((1, 2)) for e in encs[1:]])
        fx[f"W{W}"] = r
        print(tag, "W", W, {k: float(v) for k, v in losses.items()})
    torch.save(fx, os.path.join(OUT, f"loss_{tag}.pt"))
    print("wrote", f"loss_{tag}.pt")


def tokenizer_fixture(m):
    tok = m.multimodal_encoder.tokenizer
    texts = ["a man is skiing in a snowy day.", "A dog runs; the CAT sleeps!", "multimodal context omni-modal pretraining",
             "", "x " * 60]
    fx = dict(texts=texts)
    for L in (30, 77):
        o = tok(texts, padding="max_length", truncation=True, max_length=L, return_tensors="pt")
        fx[f"ids_{L}"] = o.input_ids.clone()
        fx[f"mask_{L}"] = o.attention_mask.clone()
    fx["special"] = dict(bos=tok.bos_token_id, eos=tok.eos_token_id, pad=tok.pad_token_id, mask=tok.mask_token_id)
    torch.save(fx, os.path.join(OUT, "tokenizer.pt"))
    print("wrote tokenizer.pt", fx["ids_30"][0, :12].tolist())


def vit_full_fixture():
    """Full-depth EVA01-g/14: one image, CLS + two token rows of the final LN output (GPU parity target)."""
    m = ref_import.build_mico("evaclip01_giant")
    fill(m)
    g = torch.Generator().manual_seed(99)
    x = torch.randn((1, 1, 3, 224, 224), generator=g)
    with torch.no_grad():
        out = m.forward_vision_encoder(x)
        feat = F.normalize(m.contra_head_v(m.pool_vision_for_contra(out)), dim=-1)
    torch.save(dict(rows=out[0, 0, [0, 1, 128, 256]].clone(), amax=out.abs().max(), mean=out.mean(), feat_v=feat.clone(),
                    meta=dict(input_seed=99)), os.path.join(OUT, "vit_g14_full.pt"))
    print("wrote vit_g14_full.pt")


def ckpt_fixture():
    """Reference load_from_pretrained_dir (inference_demo.py:14-116) on tiny synthetic pretrain dirs: stored (pre-rename) state
    dict + hps.json in, remapped/interpolated state dict out, for the evaclip and clip table layouts and return_modal variants."""
    import json
    import tempfile
    ref = ref_import.load()
    with ref.cwd():
        import inference_demo as ref_demo
    g = torch.Generator().manual_seed(7)
    r = lambda *s: torch.randn(*s, generator=g)
    cases = {}
    for name, vtype in (("evaclip", "evaclip01_giant"), ("clip", "clip_vit_base_16")):
        stored = {
            "video_frame_embedding": r(1, 4, 8), "audio_frame_embedding": r(1, 2, 8), "video_type_embeddings": r(1, 1, 8),
            "multimodal_encoder.bert.embeddings.word_embeddings.weight": r(6, 8).half(),
            "multimodal_encoder.cls.predictions.bias": r(6),
            "video_encoder.blocks.0.w": r(3, 3), "contra_head_t.linear.weight": r(4, 8).double(),
        }
        if name == "evaclip":
            stored["evaclip_model.visual.pos_embed"] = r(1, 1 + 9, 8)
            stored["evaclip_model.visual.patch_embed.proj.weight"] = r(8, 3, 2, 2).half()
        else:
            stored["clip_model.visual.positional_embedding"] = r(1 + 9, 8)
            stored["clip_model.visual.conv1.weight"] = r(8, 3, 2, 2)
        hps = dict(model_cfg=dict(frame_embedding_type="adaptive", max_vision_sample_num=8, max_audio_sample_num=3,
                                  vision_encoder_type=vtype, vision_resolution=10))
        d = tempfile.mkdtemp()
        os.makedirs(os.path.join(d, "ckpt")); os.makedirs(os.path.join(d, "log"))
        json.dump(hps, open(os.path.join(d, "log", "hps.json"), "w"))
        torch.save({"stale": torch.zeros(1)}, os.path.join(d, "ckpt", "model_step_9.pt"))
        torch.save(stored, os.path.join(d, "ckpt", "model_step_10.pt"))
        outs = {}
        for modal in ("full", "uni", "text"):
            ck, cfg = ref_demo.load_from_pretrained_dir(d, return_modal=modal)
            outs[modal] = {k: v.clone() for k, v in ck.items()}
        cases[name] = dict(stored=stored, hps=hps, outs=outs)
    torch.save(cases, os.path.join(OUT, "ckpt_remap.pt"))
    print("wrote ckpt_remap.pt")


def optimizer_fixture():
    """Reference AdamW (data/utils/build_optimizer.py:105-197), its parameter grouping (:11-76) and the lr schedules
    (data/utils/sched.py) on small seeded tensors: parameters after every step, moments after the last."""
    import importlib.util
    import types
    import torch.nn as nn
    for name, path in (("data", ref_import.REF_ROOT + "/data"), ("data.utils", ref_import.REF_ROOT + "/data/utils")):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [path]
            sys.modules[name] = m
    mods = {}
    for name in ("logger", "build_optimizer", "sched"):
        spec = importlib.util.spec_from_file_location("data.utils." + name, ref_import.REF_ROOT + f"/data/utils/{name}.py")
        mods[name] = importlib.util.module_from_spec(spec)
        sys.modules["data.utils." + name] = mods[name]
        spec.loader.exec_module(mods[name])
    bo, sched = mods["build_optimizer"], mods["sched"]
    g = torch.Generator().manual_seed(21)
    out = {}
    shapes = [[(5, 7), (33,)], [(4, 3, 2, 2)]]
    for cb in (True, False):
        params = [[nn.Parameter(torch.randn(*sh, generator=g)) for sh in grp] for grp in shapes]
        init = [[p.detach().clone() for p in grp] for grp in params]
        opt = bo.AdamW([dict(params=params[0], weight_decay=0.01, lr=1e-3), dict(params=params[1], weight_decay=0.0, lr=5e-4)],
                       lr=1e-3, betas=(0.9, 0.98), correct_bias=cb)
        grads, after = [], []
        for step in range(4):
            gs = [[torch.randn(*sh, generator=g) for sh in grp] for grp in shapes]
            for gi, grp in enumerate(params):
                for pi, p in enumerate(grp):
                    p.grad = None if (step == 1 and gi == 0 and pi == 1) else gs[gi][pi].clone()   # one parameter skips a step
            opt.step()
            grads.append(gs)
            after.append([[p.detach().clone() for p in grp] for grp in params])
        moments = [[(opt.state[p]["exp_avg"].clone(), opt.state[p]["exp_avg_sq"].clone(), opt.state[p]["step"]) for p in grp] for grp in params]
        out[f"correct_bias_{cb}"] = dict(init=init, grads=grads, after=after, moments=moments)
    xs = [i / 40.0 for i in range(41)]
    out["sched"] = {name: [getattr(sched, name)(x, 0.1) for x in xs] for name in ("warmup_cosine", "warmup_constant", "warmup_linear")}
    out["sched_x"] = xs
    o = types.SimpleNamespace(scheduler="warmup_linear", num_train_steps=200, warmup_ratio=0.05)
    out["get_lr_sched"] = [sched.get_lr_sched(st, o) for st in range(0, 201, 10)]

    class Tiny(nn.Module):
        def __init__(self):
            super().__init__()
            self.vision_encoder = nn.Module()
            self.vision_encoder.visual = nn.Module()
            self.vision_encoder.visual.proj = nn.Linear(3, 2)
            self.vision_encoder.visual.LayerNorm = nn.LayerNorm(2)
            self.multimodal_encoder = nn.Module()
            self.multimodal_encoder.dense = nn.Linear(2, 2)
            self.multimodal_encoder.LayerNorm = nn.LayerNorm(2)
            self.fresh_head = nn.Linear(2, 2)
            self.contra_temp = nn.Parameter(torch.tensor(0.07))

    tiny = Tiny()
    ED = ref_import.load().EasyDict
    args = ED(dict(model_cfg=dict(vision_encoder_type="evaclip01_giant"),
                   run_cfg=dict(new_params_name=["fresh_head"], weight_decay=0.01, learning_rate=1e-4, new_lr=5e-4, clip_lr=5e-7,
                                betas=[0.9, 0.98], optim="adamw")))
    ropt = bo.build_optimizer(tiny, args, None)
    ids = {id(p): n for n, p in tiny.named_parameters()}
    out["groups"] = [dict(names=[ids[id(p)] for p in gr["params"]], lr=gr["lr"], weight_decay=gr["weight_decay"], init_lr=gr["init_lr"])
                     for gr in ropt.param_groups]
    out["group_attrs"] = dict(new_params_name=ropt.new_params_name, clip_lr_visual_len=ropt.clip_lr_visual_len)
    torch.save(out, os.path.join(OUT, "optimizer.pt"))
    print("wrote optimizer.pt")


def processor_fixture():
    """Reference pre-processing logic that is its own code (not torchvision / torchaudio / decord arithmetic): `split`, the
    evaluation-mode frame / window choice, and everything AudioProcessor.__call__ does after the Kaldi filterbank (normalise,
    zero-pad, cut and pick windows; model/audioprocessor.py:45-70).  torchaudio is absent here, so its three entry points are
    stand-ins that hand a seeded filterbank to the reference code: the captured arithmetic is the reference's own."""
    import importlib.util
    import tempfile
    import types
    ta = types.ModuleType("torchaudio")
    state = {}
    ta.load = lambda f: (state["wave"], 16000)
    ta.transforms = types.SimpleNamespace(Resample=lambda a, b: (lambda w: w))
    ta.compliance = types.SimpleNamespace(kaldi=types.SimpleNamespace(fbank=lambda w, **k: state["fbank"]))
    sys.modules["torchaudio"] = ta
    spec = importlib.util.spec_from_file_location("ref_audioprocessor", ref_import.REF_ROOT + "/model/audioprocessor.py")
    ap = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ap)
    out = {"split": {}, "audio": []}
    for n, k in ((10, 4), (3, 4), (17, 8), (8, 8), (1, 3), (23, 5)):
        out["split"][(n, k)] = ap.split(list(range(n)), k)
    g = torch.Generator().manual_seed(31)
    wav = tempfile.NamedTemporaryFile(suffix=".wav")
    for T, mel, tl, sn in ((998, 32, 224, 4), (150, 32, 224, 4), (1300, 16, 224, 3), (448, 16, 224, 2)):
        fb = torch.randn(T, mel, generator=g) * 6.0 + 15.0
        state["wave"], state["fbank"] = torch.zeros(1, 16), fb
        proc = ap.AudioProcessor(melbins=mel, target_length=tl, sample_num=sn, resize_melbin_num=mel, training=False)
        res = proc(wav.name)
        out["audio"].append(dict(fbank=fb, melbins=mel, target_length=tl, sample_num=sn, out=res.clone()))
    del sys.modules["torchaudio"]
    torch.save(out, os.path.join(OUT, "processors.pt"))
    print("wrote processors.pt")


def subtitle_fixture():
    """The subtitle branch of the facade (mico.py:245-248: hidden_trans_subtitle_multimodal + subtitle type embedding) and the
    subtitle-bearing contrastive heads (contra_head_s / _vs / _vas, mico.py:385-394) of the reference, depth-1 B/16 model with the
    deterministic synthetic weights."""
    m = ref_import.build_mico("evaclip02_base", depth=1)
    fill(m)
    g = torch.Generator().manual_seed(41)
    sub_out = torch.randn(2, 7, 768, generator=g)
    pooled = {k: torch.randn(2, d, generator=g) for k, d in (("s", 768), ("vs", 768 + 768), ("vas", 768 + 768 + 768))}
    with torch.no_grad():
        out = dict(sub_in=sub_out, cond_s=m.get_multimodal_forward_input_subtitle(sub_out).clone(), pooled=pooled)
        for k in pooled:
            out["head_" + k] = getattr(m, "contra_head_" + k)(pooled[k]).clone()
    torch.save(out, os.path.join(OUT, "subtitle_b16.pt"))
    print("wrote subtitle_b16.pt")


def swin_fixture():
    """Reference model/swin.py SwinTransformer (the options data/model/general_module.py:559-576 passes) at fixture size: embed_dim 64,
    depths 2-2-2-2, heads 2-4-8-16 (head dim 32 and 7x7 windows as in Swin-B / Swin-L; both the plain and the shifted-window block of
    every stage, all three PatchMerging layers).  Output tokens, per-stage taps, gradient digests."""
    ns = ref_import.load()
    with ns.cwd():
        import model.swin as ref_swin
    torch.manual_seed(0)
    m = ref_swin.SwinTransformer(img_size=224, patch_size=4, in_chans=3, num_classes=0, embed_dim=64, depths=[2, 2, 2, 2],
                                 num_heads=[2, 4, 8, 16], window_size=7, mlp_ratio=4.0, qkv_bias=True, qk_scale=None, drop_rate=0.0,
                                 drop_path_rate=0.0, ape=False, norm_layer=torch.nn.LayerNorm, patch_norm=True, use_checkpoint=False,
                                 fused_window_process=False).eval()
    fill(m)
    g = torch.Generator().manual_seed(77)
    x = torch.randn((2, 3, 224, 224), generator=g)
    taps = []
    hooks = [layer.register_forward_hook(lambda mod, i, o: taps.append(o.detach())) for layer in m.layers]
    for p in m.parameters():
        p.requires_grad_(True)
    out = m(x)
    for h in hooks:
        h.remove()
    w = torch.randn(out.shape, generator=g) / out.numel() ** 0.5
    (out * w).sum().backward()
    named = dict(m.named_parameters())
    fx = dict(out=out.detach().clone(), tap_mean=torch.stack([t.mean() for t in taps]), tap_amax=torch.stack([t.abs().max() for t in taps]),
              tap_rows=[t[:, [0, 1, 17]].clone() for t in taps],
              grads={n: grad_digest(p.grad) for n, p in named.items() if p.grad is not None},
              meta=dict(arch="swin_tiny_test", input_seed=77))
    torch.save(fx, os.path.join(OUT, "swin_tiny.pt"))
    print("wrote swin_tiny.pt", tuple(out.shape), float(out.abs().max()), len(fx["grads"]), "gradient digests")


def masker_fixture():
    """The reference's TokenMasker itself (data/model/general_module.py:52-97, constructed as data/model/vast.py does:
    mask_token 103, random ids from [106, 30522)) run under random.seed(s) on ragged token batches: masked ids and labels draw for
    draw.  The class ends with `.cuda()` on two index tensors; there is no GPU in the build container, so Tensor.cuda is the identity
    for the duration of the call (no arithmetic involved).  `easydict` / `utils.logger` are import-time names of that file only."""
    import importlib.util
    import types
    ref_import._install_shims()
    if "utils" not in sys.modules or not hasattr(sys.modules.get("utils.logger", None), "LOGGER"):
        u = sys.modules.setdefault("utils", types.ModuleType("utils"))
        lg = types.ModuleType("utils.logger")
        lg.LOGGER = types.SimpleNamespace(info=lambda *a, **k: None, warning=lambda *a, **k: None)
        sys.modules["utils.logger"] = lg
        u.logger = lg
    spec = importlib.util.spec_from_file_location("ref_general_module", ref_import.REF_ROOT + "/data/model/general_module.py")
    gm = importlib.util.module_from_spec(spec)
    sys.dont_write_bytecode = True
    spec.loader.exec_module(gm)
    masker = gm.TokenMasker(mask_token=103, range_start=106, range_end=30522)
    cases = []
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        for seed, (b, S, p) in enumerate([(4, 12, 0.6), (8, 77, 0.6), (3, 30, 0.15), (5, 9, 0.05), (64, 77, 0.6), (2, 40, 1.0)]):
            g = torch.Generator().manual_seed(500 + seed)
            ids = torch.randint(1000, 30000, (b, S), generator=g)
            lens = torch.randint(2, S + 1, (b,), generator=g)
            lens[0] = S
            if b > 1:
                lens[1] = 2          # one maskable token: the retry loop of general_module.py:71 runs (many times at p = 0.05)
            ids[:, 0] = 101
            ids = ids * (torch.arange(S)[None] < lens[:, None])
            ids[torch.arange(b), lens - 1] = 102
            random.seed(9000 + seed)
            toks, labels = masker(ids, p)
            after = random.random()      # the generator's position afterwards: pins the NUMBER of draws as well
            cases.append(dict(seed=9000 + seed, p=p, ids=ids, masked=toks.clone(), labels=labels.clone(), next_draw=after))
    finally:
        torch.Tensor.cuda = real_cuda
    torch.save(dict(cases=cases, meta=dict(mask_token=103, range_start=106, range_end=30522,
                                           source="data/model/general_module.py:52-97 TokenMasker under random.seed")),
               os.path.join(OUT, "token_masker.pt"))
    print("wrote token_masker.pt", [(c["ids"].shape, (c["labels"] != -100).sum().item()) for c in cases])


def main(which):
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    want = lambda k: not which or k in which
    if want("vit") or want("bert") or want("facade") or want("loss") or want("tok"):
        mb = vit_fixture("evaclip02_base", 2, "b16_d2") if want("vit") else None
        if mb is None:
            mb = ref_import.build_mico("evaclip02_base", depth=2)
            fill(mb)
        if want("bert"):
            bert_fixture(mb)
        if want("tok"):
            tokenizer_fixture(mb)
        if want("facade"):
            facade_fixture(mb, "b16_d2")
        if want("loss"):
            loss_fixture(mb, "b16_d2")
        del mb
    if want("vit") or want("loss") or want("facade"):
        mg = vit_fixture("evaclip01_giant", 2, "g14_d2") if want("vit") else None
        if mg is None:
            mg = ref_import.build_mico("evaclip01_giant", depth=2)
            fill(mg)
        if want("facade"):
            facade_fixture(mg, "g14_d2")
        if want("loss"):
            loss_fixture(mg, "g14_d2")
        del mg
    if want("vitl"):     # EVA02-CLIP-L/14 (mico.py:336-340): RoPE + sub-LN + SwiGLU with the 2730-wide hidden layer (not a multiple of 8)
        del_me = vit_fixture("evaclip02_large", 2, "l14_d2")
        del del_me
    if want("vitbige"):  # EVA02-CLIP-bigE-14-plus (mico.py:341-344): the post-norm block order
        vit_bige_fixture()
    if want("swin"):
        swin_fixture()
    if want("full"):
        vit_full_fixture()
    if want("ckpt"):
        ckpt_fixture()
    if want("opt"):
        optimizer_fixture()
    if want("proc"):
        processor_fixture()
    if want("sub"):
        subtitle_fixture()
    if want("masker"):
        masker_fixture()


if __name__ == "__main__":
    main(set(sys.argv[1:]))
