"""Swin tower (reference model/swin.py): window-attention and patch-merge kernels against plain fp32 restatements,
the whole tower (forward + all parameter gradients) against the reference's own outputs (tests/golden/swin_tiny.pt), and the MiCo facade
with a Swin vision encoder (mico.py:85-86,124-126,161-163) against the oracle."""
import pytest
import torch

from common import golden, rel_err, grad_digest_check
from mico_amd import ops, runtime
from mico_amd.model import MiCo, default_cfg
from mico_amd.model.swin import SWIN_CONFIGS, SwinTransformer
from mico_amd.weights import synth_inputs, synth_state_dict
from oracle import mico_oracle as O

pytestmark = pytest.mark.gpu


def _window_attention_ref(qkv, table, B, res, heads, shift):
    """swin.py:258-289 + :134-152 on [B*L, 3C] fp32 tokens: roll, partition, attention with bias (+ mask), reverse, roll back."""
    C = qkv.shape[1] // 3
    x = qkv.view(B, res, res, 3 * C)
    if shift:
        x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
    xw = O.swin_window_partition(x, 7).view(-1, 49, 3, heads, 32).permute(2, 0, 3, 1, 4)
    q, k, v = xw[0] * 32 ** -0.5, xw[1], xw[2]
    attn = q @ k.transpose(-2, -1) + table[O.swin_rel_index(7).view(-1).to(table.device)].view(49, 49, heads).permute(2, 0, 1).unsqueeze(0)
    if shift:
        mask = O.swin_shift_mask(res, 7, shift).to(qkv.device)
        nW = mask.shape[0]
        attn = (attn.view(-1, nW, heads, 49, 49) + mask.unsqueeze(1).unsqueeze(0)).view(-1, heads, 49, 49)
    o = (attn.softmax(-1) @ v).transpose(1, 2).reshape(-1, 7, 7, C)
    o = O.swin_window_reverse(o, 7, res, res)
    if shift:
        o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))
    return o.reshape(B * res * res, C)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,res,heads,shift", [(3, 14, 2, 0), (3, 14, 2, 3), (2, 28, 4, 3), (5, 7, 3, 0), (70, 14, 1, 3)])
def test_window_attention(cuda, dtype, B, res, heads, shift):
    g = torch.Generator().manual_seed(res * 10 + shift)
    C = heads * 32
    qkv16 = torch.randn((B * res * res, 3 * C), generator=g).to(cuda).to(dtype)
    table = (0.5 * torch.randn((169, heads), generator=g)).to(cuda)
    dout16 = torch.randn((B * res * res, C), generator=g).to(cuda).to(dtype)
    out = torch.empty((B * res * res, C), device=cuda, dtype=dtype)
    lse = torch.empty((B * res * res, heads), device=cuda)
    ops.win_attn_fwd(qkv16, out, lse, table, B, res, heads, shift, 32 ** -0.5)
    dqkv = torch.empty_like(qkv16)
    dtab = torch.zeros_like(table)
    ops.win_attn_bwd(qkv16, dout16, lse, table, dqkv, dtab, B, res, heads, shift, 32 ** -0.5, dbias_scale=0.5)
    q32 = qkv16.float().requires_grad_(True)
    t32 = table.clone().requires_grad_(True)
    ref = _window_attention_ref(q32, t32, B, res, heads, shift)
    ref.backward(dout16.float())
    tol = 2e-3 if dtype == torch.float16 else 1.6e-2          # 16-bit rounding of the stored outputs; the arithmetic is fp32
    assert rel_err(out, ref) < tol
    assert rel_err(dqkv, q32.grad) < tol
    assert rel_err(dtab, 0.5 * t32.grad) < 1e-4


@pytest.mark.parametrize("B,res,C", [(2, 56, 64), (3, 14, 512), (1, 28, 192)])
def test_patch_merge(cuda, B, res, C):
    x = torch.randn((B, res * res, C), device=cuda)
    out = torch.empty((B, res * res // 4, 4 * C), device=cuda)
    ops.patch_merge(x, out, B, res, C)
    v = x.view(B, res, res, C)
    ref = torch.cat([v[:, 0::2, 0::2], v[:, 1::2, 0::2], v[:, 0::2, 1::2], v[:, 1::2, 1::2]], -1).view(B, -1, 4 * C)   # swin.py:340-346
    assert torch.equal(out, ref)
    back = torch.empty_like(x)
    ops.patch_merge(out, back, B, res, C, backward=True)
    assert torch.equal(back, x)


def _tower(cuda, drop_path_rate=0.0):
    c = SWIN_CONFIGS["swin_tiny_test"]
    m = SwinTransformer(embed_dim=c["embed_dim"], depths=c["depths"], num_heads=c["num_heads"], drop_path_rate=drop_path_rate)
    sd = synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()})
    m.load_state_dict(sd, strict=False)
    return m.to(cuda).eval(), sd


@pytest.mark.parametrize("dtype,ftol,gtol", [(torch.float16, 1e-3, 2e-2), (torch.bfloat16, 1.2e-2, 6e-2)])
def test_swin_tower_vs_reference(cuda, dtype, ftol, gtol):
    """Forward tokens and every parameter gradient of the fixture-sized tower against the reference's own run (fp16 in the tests' parity
    configuration - split-precision GEMMs - under the 1e-3 gate, bf16 reported at its own tolerance, as for the EVA towers)."""
    m, sd = _tower(cuda)
    fx = golden("swin_tiny.pt")
    g = torch.Generator().manual_seed(fx["meta"]["input_seed"])
    x = torch.randn((2, 3, 224, 224), generator=g)
    w = torch.randn(fx["out"].shape, generator=g) / fx["out"].numel() ** 0.5
    with runtime.precision(dtype):
        out = m(x.to(cuda))
        assert out.shape == fx["out"].shape
        e = rel_err(out, fx["out"])
        (out * w.to(cuda)).sum().backward()
    named = dict(m.named_parameters())
    errs = {n: grad_digest_check(d, named[n].grad, None) for n, d in fx["grads"].items()}
    worst = max(errs, key=errs.get)
    print(f"swin_tiny {dtype}: fwd {e:.2e}  worst grad {errs[worst]:.2e} ({worst})")
    assert e < ftol
    assert errs[worst] < gtol, (worst, errs[worst])


def test_swin_drop_path_injected(cuda):
    """Stochastic depth with injected per-sample multipliers (0 or 1/keep) against the oracle evaluating the same masks."""
    m, sd = _tower(cuda)
    g = torch.Generator().manual_seed(3)
    x = torch.randn((3, 3, 224, 224), generator=g)
    keep = 0.7
    dps = (torch.bernoulli(torch.full((8, 2, 3), keep), generator=g) / keep).contiguous()
    dps[0, 0, 1] = 0.0
    ref = O.swin_forward(sd, x, O.SWIN_ARCHS["swin_tiny_test"], pre="", drop_path_scale=dps)
    with runtime.precision(torch.float16), torch.no_grad():
        out = m.forward_features(x.to(cuda), drop_path_scale=dps)
    assert rel_err(out, ref) < 1e-3


def test_mico_with_swin_encoder(cuda):
    """The facade branches the reference takes for a Swin vision encoder (mico.py:124-126 forward, :139-140 audio as three identical
    channels, :161-163 token-mean pooling) against the oracle, and one alignment step (ITC + ITM + CAP) forward + backward through it."""
    torch.manual_seed(0)
    m = MiCo(default_cfg("swin_tiny_test"))
    sd = synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()})
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected
    m = m.to(cuda).eval()
    assert m.vision_dim == 512
    inp = synth_inputs(dict(b=2, vision=2, audio=1, S=12), seed=9)
    arch = O.SWIN_ARCHS["swin_tiny_test"]
    with runtime.precision(torch.float16), torch.no_grad():
        tv = m.forward_vision_encoder(inp["vision_pixels"].to(cuda))
        ta = m.forward_audio_encoder(inp["audio_spectrograms"].to(cuda))
        fv = m.pool_vision_for_contra(tv)
        cv = m.get_multimodal_forward_input_vision(tv)
    rv = O.swin_forward(sd, inp["vision_pixels"].reshape(4, 3, 224, 224), arch).reshape(2, 2, 49, 512)
    ra = O.swin_forward(sd, inp["audio_spectrograms"].reshape(2, 1, 224, 224).repeat(1, 3, 1, 1), arch).reshape(2, 1, 49, 512)
    assert tv.shape == rv.shape and rel_err(tv, rv) < 1e-3
    assert ta.shape == ra.shape and rel_err(ta, ra) < 1e-3
    assert rel_err(fv, rv.mean(dim=2).mean(dim=1)) < 1e-3
    assert rel_err(cv, O.multimodal_input(sd, "vision", rv)) < 1e-3
    m.train()
    batch = {k: v.to(cuda) for k, v in inp.items()}
    with runtime.precision(torch.float16):
        out = m(batch, "ret%tva_cap%tva", compute_loss=True)
        loss = sum(v for k, v in out.items() if k.startswith("loss"))
        loss.backward()
    assert torch.isfinite(loss)
    gtab = m.vision_encoder.layers[0].blocks[1].attn.relative_position_bias_table.grad
    gpe = m.vision_encoder.patch_embed.proj.weight.grad
    assert gtab is not None and gpe is not None and torch.isfinite(gtab).all() and gtab.abs().sum() > 0 and gpe.abs().sum() > 0


@pytest.mark.parametrize("name,B", [("swin_base_22k_224", 4), ("swin_large_22k_224", 2)])
def test_swin_full_size_vs_oracle(cuda, name, B):
    """Swin-B / Swin-L at the sizes data/model/general_module.py:532-539 selects (24 blocks, 1024 / 1536 output channels; Swin-L's last
    PatchMerging norm is the 3072-column LayerNorm) against the oracle: tokens, and parameter gradients of the first / last block, the patch
    embedding, one PatchMerging layer and a shifted block's bias table through autograd on the oracle."""
    torch.set_num_threads(32)
    c = SWIN_CONFIGS[name]
    m = SwinTransformer(embed_dim=c["embed_dim"], depths=c["depths"], num_heads=c["num_heads"], drop_path_rate=0.0)
    sd = synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()})
    m.load_state_dict(sd, strict=False)
    m = m.to(cuda).eval()
    g = torch.Generator().manual_seed(11)
    x = torch.randn((B, 3, 224, 224), generator=g)
    w = torch.randn((B, 49, m.num_features), generator=g) / (B * 49 * m.num_features) ** 0.5
    watch = ["patch_embed.proj.weight", "layers.0.blocks.0.attn.qkv.weight", "layers.0.blocks.1.attn.relative_position_bias_table",
             "layers.1.downsample.reduction.weight", "layers.2.downsample.norm.weight", "layers.2.blocks.17.mlp.fc1.weight",
             "layers.3.blocks.1.attn.proj.bias", "norm.weight"]
    sdo = {k: (v.clone().requires_grad_(True) if k in watch else v) for k, v in sd.items()}
    ref = O.swin_forward(sdo, x, O.SWIN_ARCHS[name], pre="")
    (ref * w).sum().backward()
    with runtime.precision(torch.float16):
        out = m(x.to(cuda))
        e = rel_err(out, ref.detach())
        (out * w.to(cuda)).sum().backward()
    named = dict(m.named_parameters())
    ge = {n: rel_err(named[n].grad, sdo[n].grad) for n in watch}
    print(f"{name}: fwd {e:.2e}  grads {max(ge.values()):.2e}")
    assert e < 1e-3
    assert max(ge.values()) < 2e-2, ge
