"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol include/mico_hip.h declares, the host
modules keep the reference's state-dict surface, the tokenizer reproduces the reference's token ids bit-exactly, host
logic (token masking, checkpoint remap, task grammar) behaves as the reference's, and nothing silently computes on CPU."""
import os
import re

import pytest
import torch

from common import golden, build_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "mico_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(mico_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    import ctypes
    from mico_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "libmico_hip.so has not been built (python -c 'import __graft_entry__ as g; g.build()')"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = _declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/mico_hip.h but not exported"
    # and the ctypes prototype table covers exactly the header
    assert sorted(_lib.PROTOTYPES) == syms
    l = _lib.lib()
    assert l.mico_version() == _lib.ABI_VERSION     # lib() itself refuses any other version


def test_ctypes_structs_match_the_compiled_layout():
    """The parameter structs the ctypes binding builds are the structs the library was compiled with: size and every field offset,
    in declaration order (a stray trailing field in the ctypes GemmEpilogue went unnoticed because only symbol names
    were compared).  No kernel runs: mico_struct_layout() is host code."""
    import ctypes
    from mico_amd import _lib
    l = _lib.lib()
    n = l.mico_struct_layout(None, 0)
    buf = (ctypes.c_int * n)()
    assert l.mico_struct_layout(buf, n) == n
    table, cur = [], []
    for v in buf:
        if v == -1:
            table.append(cur)
            cur = []
        else:
            cur.append(v)
    assert len(table) == 4
    for cls, (size, *offs) in zip((_lib.GemmEpilogue, _lib.AttnParams, _lib.LnFwdParams, _lib.LnBwdParams), table):
        assert ctypes.sizeof(cls) == size, (cls.__name__, ctypes.sizeof(cls), size)
        mine = [getattr(cls, name).offset for name, _ in cls._fields_]
        assert mine == offs, (cls.__name__, mine, offs)
    # and the header's field lists are what the table was built from (a field added to the header but not to the table would hide here)
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "mico_hip.h")).read(), flags=re.S)
    for struct, cls in (("mico_gemm_epilogue", _lib.GemmEpilogue), ("mico_attn_params", _lib.AttnParams),
                        ("mico_ln_fwd_params", _lib.LnFwdParams), ("mico_ln_bwd_params", _lib.LnBwdParams)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), hdr, re.S).group(1)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            first, *rest = decl.split(",")
            names.append(re.findall(r"([A-Za-z_][A-Za-z0-9_]*)\s*(?:\[\d+\])?$", first.strip())[0])
            names += [re.findall(r"([A-Za-z_][A-Za-z0-9_]*)", r.strip())[0] for r in rest]
        assert names == [n_ for n_, _ in cls._fields_], (struct, names)


def test_missing_library_fails_loudly(monkeypatch):
    from mico_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libmico_hip.so")
    with pytest.raises(_lib.MicoHipError):
        _lib.lib()


def test_no_cpu_compute_path():
    from mico_amd._lib import MicoHipError
    m, _ = build_model("evaclip02_base", 1)
    with pytest.raises(MicoHipError):
        m.forward_vision_encoder(torch.zeros(1, 1, 3, 224, 224))
    with pytest.raises(MicoHipError):
        m.forward_multimodal_encoder(torch.ones(1, 4, dtype=torch.long), torch.ones(1, 4, dtype=torch.long))


def test_product_never_imports_oracle():
    for dp, _, files in os.walk(os.path.join(ROOT, "mico_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S).replace("# oracle", ""), f


@pytest.mark.parametrize("vtype", ["evaclip02_base", "evaclip01_giant"])
def test_state_dict_surface_matches_reference(vtype):
    """Every PARAMETER key of the reference MiCo (recorded from the reference in tests/golden/state_dict_keys.pt) exists
    with the same shape, so reference checkpoints load key-for-key."""
    from mico_amd.model import MiCo, default_cfg
    ref = golden("state_dict_keys.pt")[vtype]
    try:      # names and shapes only: the 1.2 B parameters of g/14 need no storage (35-70 s of CPU initialisation otherwise)
        with torch.device("meta"):
            m = MiCo(default_cfg(vtype, vision_layers=None))
    except Exception:
        m = MiCo(default_cfg(vtype, vision_layers=None))
    mine = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    ref_params = {k: s for k, (s, is_param) in ref.items() if is_param}
    missing = [k for k in ref_params if k not in mine]
    assert not missing, missing[:10]
    bad = [k for k, s in ref_params.items() if mine[k] != s]
    assert not bad, bad[:10]
    extra = [k for k in mine if k not in ref]
    assert not extra, extra[:10]
    assert sum(p.numel() for p in m.parameters()) == {"evaclip02_base": 256067390 - 30522 * 768, "evaclip01_giant": 1187144638 - 30522 * 768}[vtype] \
        or sum(p.numel() for p in m.parameters()) in (256067390, 1187144638)


def test_tokenizer_ids_bit_exact():
    from mico_amd.model import build_tokenizer
    fx = golden("tokenizer.pt")
    tok = build_tokenizer()
    assert dict(bos=tok.bos_token_id, eos=tok.eos_token_id, pad=tok.pad_token_id, mask=tok.mask_token_id) == fx["special"] == \
        dict(bos=101, eos=102, pad=0, mask=103)
    for L in (30, 77):
        o = tok(fx["texts"], padding="max_length", truncation=True, max_length=L, return_tensors="pt")
        assert torch.equal(o.input_ids, fx[f"ids_{L}"])
        assert torch.equal(o.attention_mask, fx[f"mask_{L}"])


def test_token_masker_rules():
    import random
    from mico_amd.model import TokenMasker
    from oracle.mico_oracle import token_masker
    ids = torch.tensor([[101, 2000, 2001, 2002, 2003, 102, 0, 0], [101, 5, 102, 0, 0, 0, 0, 0]])
    tm = TokenMasker(rng=random.Random(7))
    toks, labels = tm(ids, 0.6)
    ref_t, ref_l = token_masker(ids, 0.6, random.Random(7))
    assert torch.equal(toks, ref_t) and torch.equal(labels, ref_l)   # same draws -> same tokens as the restated reference rule
    assert (labels[:, 0] == -100).all() and (labels[ids == 0] == -100).all() and ((labels != -100).sum(1) >= 1).all()


def test_token_masker_host_path_equals_reference_class():
    """The product's host TokenMasker (CPU tokens / injected random.Random) against the reference class's own output under the same seed
    (tests/golden/token_masker.pt, oracle/make_golden.py `masker`): ids, labels and the generator position afterwards."""
    import random
    from common import golden
    from mico_amd.model import TokenMasker
    fx = golden("token_masker.pt")
    for c in fx["cases"]:
        rng = random.Random(c["seed"])
        toks, labels = TokenMasker(rng=rng)(c["ids"], c["p"])
        assert torch.equal(toks, c["masked"]) and torch.equal(labels, c["labels"]), c["seed"]
        assert rng.random() == c["next_draw"]


def test_modify_checkpoint_remap_and_interpolation():
    import torch.nn.functional as F
    from mico_amd.model import MiCo, default_cfg
    m = MiCo(default_cfg("evaclip02_base", vision_layers=1, vision_resolution=224, max_vision_sample_num=4))
    g = torch.Generator().manual_seed(0)
    ck = {"video_frame_embedding": torch.randn(1, 8, 768, generator=g), "audio_frame_embedding": torch.randn(1, 2, 768, generator=g),
          "evaclip_model.visual.pos_embed": torch.randn(1, 1 + 12 * 12, 768, generator=g),
          "evaclip_model.visual.patch_embed.proj.weight": torch.randn(768, 3, 16, 16, generator=g),
          "contra_temp": torch.tensor(0.05, dtype=torch.float64)}
    out = m.modify_checkpoint(dict(ck))
    assert out["vision_frame_embedding"].shape == (1, 4, 768)
    assert torch.equal(out["vision_frame_embedding"], F.interpolate(ck["video_frame_embedding"].permute(0, 2, 1), 4, mode="nearest").permute(0, 2, 1))
    assert out["audio_frame_embedding"].shape == (1, 4, 768)
    assert out["vision_encoder.visual.pos_embed"].shape == (1, 197, 768)
    assert torch.equal(out["vision_encoder.visual.pos_embed"][0, 0], ck["evaclip_model.visual.pos_embed"][0, 0])
    assert out["contra_temp"].dtype == torch.float32


def test_unknown_encoder_type_raises():
    from mico_amd.model import MiCo, default_cfg
    for t in ("videoswin_base_k600_22k", "clip_vit_base_16", "swin_small_1k"):   # mico.py:83-90 accepts none of these here
        with pytest.raises(NotImplementedError):
            MiCo(default_cfg(t))


def test_frame_embedding_nearest_index_matches_interpolate():
    import torch.nn.functional as F
    for src, n in [(8, 3), (4, 2), (8, 5), (1, 4), (4, 8)]:
        fe = torch.randn(1, src, 16)
        idx = torch.floor(torch.arange(n, dtype=torch.float32) * (src / n)).long()
        assert torch.equal(fe[:, idx], F.interpolate(fe.permute(0, 2, 1), n, mode="nearest").permute(0, 2, 1))


def test_bench_refuses_more_gpus_than_visible():
    """`python bench.py --gpus N` outside torchrun launches its own N ranks - and says so loudly when the box does not have N GPUs,
    instead of silently measuring one rank (`assert world == args.gpus or world == 1`)."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "--gpus 2 but only" in (r.stderr + r.stdout), (r.returncode, r.stderr[-400:])
    # under a launcher whose world size disagrees with --gpus: refused as well
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout), (r.returncode, r.stderr[-400:])


def test_drop_plan_transitions_and_grad_arena_groups():
    """Host logic behind two fused backward paths: DropPlan.transition (where each kept frame of a branch sits in the PREVIOUS branch's
    compact list - the LayerNorm backward's dx16_dst - and which frames only that branch keeps) against a brute-force restatement; GradArena
    groups (BERT's q | k | v gradient views adjacent, so the fused projection's weight-gradient GEMM accumulates into one matrix view)."""
    from mico_amd.functional import DropPlan, GradArena
    torch.manual_seed(0)
    depth, Bf = 4, 11
    sc = (torch.rand(depth, 2, Bf) > 0.35).float() * 1.25
    sc[1, 0] = 1.25            # a branch that keeps everything
    sc[2, 1] = 0.0             # and one that keeps nothing
    stats = list(DropPlan.stats)
    plan = DropPlan(sc, Bf, torch.device("cpu"))
    DropPlan.stats[:] = stats
    keep = (sc != 0).reshape(-1, Bf)
    assert plan.transition(0, 0) is None
    for j in range(1, 2 * depth):
        dst, only, only_dst = plan.transition(j // 2, j % 2)
        fx = keep[j].nonzero()[:, 0].tolist()
        fy = keep[j - 1].nonzero()[:, 0].tolist()
        assert dst.tolist() == [fy.index(f) if f in fy else -1 for f in fx]
        assert only.tolist() == [f for f in fy if f not in fx]
        assert only_dst.tolist() == [fy.index(f) for f in only.tolist()]
        # together they cover the previous branch's compact list exactly once
        assert sorted([d for d in dst.tolist() if d >= 0] + only_dst.tolist()) == list(range(len(fy)))
    ps = [torch.zeros(8, 4), torch.zeros(8), torch.zeros(8, 4), torch.zeros(8), torch.zeros(8, 4), torch.zeros(8), torch.zeros(5)]
    ar = GradArena(ps, [[0, 2, 4], [1, 3, 5]])
    ar.fused([0, 2, 4], (24, 4)).copy_(torch.arange(96.0).view(24, 4))
    ar.fused([1, 3, 5], (24,)).fill_(7.0)
    assert torch.equal(ar.get(2), torch.arange(32.0, 64.0).view(8, 4)) and torch.equal(ar.get(5), torch.full((8,), 7.0))
    assert ar.get(6).abs().sum() == 0 and ar.views[6].shape == (5,)
    plain = GradArena(ps)
    assert plain.offsets == [0, 32, 40, 72, 80, 112, 120]        # registration order (the towers' block spans rely on it)
    with pytest.raises(AssertionError):
        plain.fused([0, 2], (12, 4))


def test_shared_grad_arena_sessions():
    """functional.GradArena.session: several autograd nodes over the same parameters accumulate in ONE arena per backward pass; a view is handed
    to autograd by the first node that touched the parameter; untouched parameters keep grad None; a parameter with another consumer is private
    (per-node buffers, summed by autograd) - against plain autograd, with the sharing on and off, twice in a row (a new arena per pass)."""
    import torch
    from mico_amd import runtime
    from mico_amd.functional import GradArena

    class Node(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, use_b, *params):
            w, bias, unused, tied = params
            ctx.save_for_backward(x)
            ctx.params, ctx.use_b = params, use_b
            y = x @ w.detach() + tied.detach().sum()
            return y + bias.detach() if use_b else y

        @staticmethod
        def backward(ctx, g):
            (x,) = ctx.saved_tensors
            grads = GradArena.session(ctx.params, private=(3,))
            grads.get(0).add_(x.t() @ g)                 # kernels accumulate into the arena views
            if ctx.use_b:
                grads.get(1).add_(g.sum(0))
            grads.get(3).add_(g.sum())
            return (None, None) + grads.result()

    torch.manual_seed(0)
    xs = [torch.randn(5, 4) for _ in range(3)]
    ref = None
    for share in (False, True, True):
        runtime.CFG.share_grad_arena = share
        try:
            w, bias, unused, tied = (torch.nn.Parameter(torch.randn(*s)) for s in ((4, 3), (3,), (2,), (6,)))
            torch.manual_seed(1)
            for p, v in zip((w, bias, unused, tied), (torch.randn(4, 3), torch.randn(3), torch.randn(2), torch.randn(6))):
                p.data.copy_(v)
            # the first node does not touch `bias`, the later ones do; `tied` has a consumer outside the nodes
            loss = sum((Node.apply(x, i > 0, w, bias, unused, tied) ** 2).sum() for i, x in enumerate(xs)) + (tied * 3.0).sum()
            loss.backward()
            got = [None if p.grad is None else p.grad.clone() for p in (w, bias, unused, tied)]
        finally:
            runtime.CFG.share_grad_arena = True
        assert got[2] is None                                # nobody touched it
        if ref is None:
            ref = got
        else:
            for a, b in zip(ref, got):
                assert (a is None) == (b is None) and (a is None or torch.allclose(a, b, rtol=1e-6, atol=1e-6))
    assert len(GradArena._shared) <= 1


def test_grad_arena_private_fused_group_with_outside_producer():
    """a parameter pair written as ONE fused view by the arena nodes (BERT's cross-attention key | value gradients) that ALSO
    receives a gradient from a producer outside those nodes (CrossKVFn) - in every order of arrival at the parameter's accumulator.  As part of
    the shared arena the second defined gradient makes the engine replace the accumulator and later in-place additions are lost; as a private
    group (a fused buffer per node) autograd sums everything.  Checked against plain autograd."""
    import torch
    from mico_amd.functional import GradArena

    class Node(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, private, *params):
            k, v = params
            ctx.save_for_backward(x)
            ctx.params, ctx.private = params, private
            return x @ torch.cat((k.detach(), v.detach()), 0).t()

        @staticmethod
        def backward(ctx, g):
            (x,) = ctx.saved_tensors
            grads = GradArena.session(ctx.params, groups=[[0, 1]], private=(0, 1) if ctx.private else ())
            grads.fused([0, 1], (8, 4)).add_(g.t() @ x)        # the fused weight-gradient GEMM accumulates into the adjacent views
            return (None, None) + grads.result()

    class Outside(torch.autograd.Function):                    # the CrossKVFn-like producer: defined gradients of its own for k and v
        @staticmethod
        def forward(ctx, x, k, v):
            ctx.save_for_backward(x)
            return x @ k.detach().t() + x @ v.detach().t()

        @staticmethod
        def backward(ctx, g):
            (x,) = ctx.saved_tensors
            return None, g.t() @ x, g.t() @ x

    torch.manual_seed(0)
    xs = [torch.randn(5, 4) for _ in range(4)]
    k0, v0 = torch.randn(4, 4), torch.randn(4, 4)

    def run(order, private, plain=False):
        k, v = torch.nn.Parameter(k0.clone()), torch.nn.Parameter(v0.clone())
        terms = []
        for kind, x in zip(order, xs):
            if kind == "n":
                y = x @ torch.cat((k, v), 0).t() if plain else Node.apply(x, private, k, v)
            else:
                y = x @ k.t() + x @ v.t() if plain else Outside.apply(x, k, v)
            terms.append((y ** 2).sum())
        sum(terms).backward()
        return k.grad.clone(), v.grad.clone()

    for order in ("nnon", "onnn", "nnno", "nonn"):
        ref = run(order, True, plain=True)
        got = run(order, True)
        for a, b in zip(ref, got):
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-5), order


def test_dkv_session_contract_under_the_autograd_engine():
    """functional.DkvSession's contract with the engine, restated with toy nodes on the CPU (the real readers are BertFn passes on the GPU:
    tests/test_model_gpu.py::test_shared_cross_kv_equals_per_pass_projection): several readers of ONE producer output keep one gradient buffer - the
    first backward writes and registers it and hands it to autograd, later ones add IN PLACE and return None, the producer's backward sees the
    complete sum (the engine keeps a single incoming gradient by reference, it does not copy it) and ends the session.  A reader that cannot take
    part returns its own buffer and closes the session: the readers after it fall back to buffers of their own, autograd sums, nothing is lost.
    Every order of readers, against plain autograd."""
    import torch
    from mico_amd.functional import DkvSession

    seen = {}

    class Producer(torch.autograd.Function):                     # CrossKVFn: y = x w, session ended by the backward
        @staticmethod
        def forward(ctx, session, x, w):
            ctx.session, ctx.x, ctx.w = session, x.detach(), w.detach()
            return x.detach() @ w.detach()

        @staticmethod
        def backward(ctx, g):
            seen["own_at_producer"] = ctx.session.own
            ctx.session.reset()
            return None, g @ ctx.w.t(), ctx.x.t() @ g

    class Reader(torch.autograd.Function):                       # a BertFn pass reading kv_own
        @staticmethod
        def forward(ctx, kv, scale, can_add):
            ctx.session, ctx.scale, ctx.can_add = getattr(kv, "_mico_dkv", None), scale, can_add
            return kv.detach() * scale

        @staticmethod
        def backward(ctx, g):
            sess = ctx.session if ctx.needs_input_grad[0] else None
            mine = g * ctx.scale
            if sess is not None and sess.own is not None and not sess.closed:
                if ctx.can_add:
                    sess.own.add_(mine)                          # (the attention kernel's dkv_accumulate)
                    DkvSession.accumulated += 1
                    return None, None, None
                sess.closed = True
            buf = mine.clone()
            if sess is not None:
                if sess.own is None and not sess.closed:
                    sess.own = buf
                else:
                    sess.closed = True
            return buf, None, None

    torch.manual_seed(1)
    x0, w0 = torch.randn(6, 3), torch.randn(3, 5)
    for can in ((True, True, True), (True, False, True), (False, True, True), (True, True, False)):
        ref = None
        for plain in (True, False):
            x, w = x0.clone().requires_grad_(), torch.nn.Parameter(w0.clone())
            sess = DkvSession()
            kv = x @ w if plain else Producer.apply(sess, x, w)
            if not plain:
                kv._mico_dkv = sess
            a0 = DkvSession.accumulated
            terms = [((kv * s if plain else Reader.apply(kv, s, c)) ** 2).sum() for s, c in zip((1.0, 2.0, 3.0), can)]
            sum(terms).backward()
            got = (x.grad.clone(), w.grad.clone())
            if plain:
                ref = got
                continue
            assert sess.own is None and not sess.closed                   # ended by the producer's backward
            # readers run in reverse creation order: every reader that may add, and runs before a non-adding one closes the session, adds
            expect = {(True, True, True): 2, (True, False, True): 0, (False, True, True): 1, (True, True, False): 2}[can]
            assert DkvSession.accumulated - a0 == expect, (can, DkvSession.accumulated - a0)
            for r, g in zip(ref, got):
                assert torch.allclose(r, g, rtol=1e-5, atol=1e-5), can


def test_bench_compact_line_keeps_the_contract():
    """bench.py's stdout line is the compact form of the full record (round 6: the round-5 line was > 8 KB and a driver's tail cut its head off):
    every contract field verbatim, `roofline` with the step-level figures, `cpu_baseline`, and it fits a tail.  Input: the committed full record of
    the round's collection (profiles/r06_bench_full.json)."""
    import json
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    full = json.load(open(os.path.join(root, "profiles", "r06_bench_full.json")))
    line = bench.compact_line(full)
    txt = json.dumps(line)
    assert len(txt) < 8192
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert line[k] == full[k], k
    assert line["config"]["workload"] == full["config"]["workload"] and "model" not in line["config"]
    r = line["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["step_frac"] == full["step_mfma_frac"] and r["executed_tflop_per_sample"] == full["tflop_per_sample"]["executed"]
    assert abs(r["step_frac"] - r["executed_tflop_per_sample"] * full["value"] / 2500.0) < 1e-6      # the step figure is recomputable from the line
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(line["cpu_baseline"]) and line["cpu_baseline"]["kind"] == "port"
    assert line["parity"]["worst"] == full["parity"]["worst"] < line["parity"]["gate"]
    assert "configs2_img_aud_txt_bf16" in line["secondary"]
