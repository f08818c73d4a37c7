"""inference_demo.py flow (BASELINE configs[0] plumbing): synthetic pretrain dir with pre-rename keys -> load_from_pretrained_dir
-> MiCo.from_pretrained -> ImageProcessor on a synthetic 428x640 jpeg -> image/text features, similarity, ITM score, beam-search
caption, against the CPU oracle evaluated on the same processed pixels and the same (remapped) weights.  Token ids bit-exact."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from common import rel_err

pytestmark = pytest.mark.gpu


def test_demo_flow(cuda, tmp_path):
    import inference_demo as demo
    from mico_amd import runtime
    from mico_amd.model import MiCo
    from mico_amd.model.imageprocessor import ImageProcessor
    from oracle import mico_oracle as O
    from PIL import Image
    rng = np.random.RandomState(0)
    img = (rng.rand(428, 640, 3) * 255).astype(np.uint8)
    path = str(tmp_path / "test.jpeg")
    Image.fromarray(img).save(path, quality=95)
    pdir = str(tmp_path / "MiCo-synth")
    cfg, sd = demo.write_synthetic_pretrain_dir(pdir, "evaclip02_base", steps=(3, 12), vision_layers=2, max_vision_sample_num=8)
    ckpt, opts = demo.load_from_pretrained_dir(pdir)
    assert set(ckpt) == set(sd)                      # every key renamed back
    for k in sd:
        assert torch.equal(ckpt[k].float(), sd[k].float()), k
    runtime.set_compute_dtype(torch.float16)
    model = MiCo.from_pretrained(opts, ckpt).to(cuda).eval()
    proc = ImageProcessor(224, "swin", training=True)
    x = proc(path)
    assert x.shape == (1, 3, 224, 224)
    texts = ["a man is skiing in a snowy day.", "it's a hot day"]
    out = demo.run_demo(model, x, texts, cuda)
    assert out["input_ids"][0, :11].tolist() == [101, 1037, 2158, 2003, 12701, 1999, 1037, 20981, 2154, 1012, 102]
    # oracle on the same pixels
    sdo = dict(sd)
    sdo["multimodal_encoder.cls.predictions.decoder.weight"] = sdo["multimodal_encoder.bert.embeddings.word_embeddings.weight"]
    arch = O.ARCHS["evaclip02_base"]
    with torch.no_grad():
        vo = O.forward_vision_encoder(sdo, arch, x.unsqueeze(1))
        fv = O.contra_feat(sdo, "contra_head_v", O.pool_for_contra(vo))
        ids = out["input_ids"].cpu()
        am = (ids != 0).long()
        seq = O.bert_forward(sdo, ids, am)
        ft = O.contra_feat(sdo, "contra_head_t", seq[:, 0])
        cond = O.multimodal_input(sdo, "vision", vo).expand(2, -1, -1)
        sc = F.softmax(O.itm_head(sdo, O.bert_forward(sdo, ids, am, cond)[:, 0]), dim=1)[:, 1]
    assert rel_err(out["feat_v"], fv) < 1e-3 and rel_err(out["feat_t"], ft) < 1e-3
    assert (out["sim_t2v"].cpu() - ft @ fv.t()).abs().max() < 1e-3
    assert rel_err(out["itm_scores"], sc) < 2e-3
    assert model.vision_encoder.text is None
    # caption step: same beam search on the oracle's condition tensor, token ids bit-exact
    with torch.no_grad():
        cap_ref = O.generate_beam(sdo, O.multimodal_input(sdo, "vision", vo), model.max_caption_len, model.beam_size, 0.6)
    assert out["caption_ids"].cpu().tolist() == cap_ref.tolist()
    assert len(out["captions"]) == 1 and isinstance(out["captions"][0], str)
