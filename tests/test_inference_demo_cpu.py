"""load_from_pretrained_dir / ImageProcessor host logic (no GPU): latest-step selection, key renames, frame-embedding and
position-table interpolation, return_modal variants, mean/std selection, error behaviour (None for unreadable files)."""
import json
import os

import numpy as np
import torch


def test_loader_and_processor(tmp_path):
    import inference_demo as demo
    from mico_amd.model.imageprocessor import ImageProcessor
    pdir = str(tmp_path / "p")
    cfg, sd = demo.write_synthetic_pretrain_dir(pdir, "evaclip02_base", steps=(3, 12), vision_layers=1)
    ckpt, opts = demo.load_from_pretrained_dir(pdir)
    assert set(ckpt) == set(sd) and all(torch.equal(ckpt[k].float(), sd[k].float()) for k in sd)
    assert not any("video" in k or "evaclip_model" in k for k in ckpt)
    # interpolation paths: ask for another sample number and resolution through hps.json
    h = json.load(open(os.path.join(pdir, "log", "hps.json")))
    h["model_cfg"]["max_vision_sample_num"] = 2 * cfg.max_vision_sample_num
    h["model_cfg"]["vision_resolution"] = 448
    json.dump(h, open(os.path.join(pdir, "log", "hps.json"), "w"))
    ck2, o2 = demo.load_from_pretrained_dir(pdir)
    fe, fe2 = ckpt["vision_frame_embedding"], ck2["vision_frame_embedding"]
    assert fe2.shape[1] == 2 * fe.shape[1] and torch.equal(fe2[:, ::2], fe) and torch.equal(fe2[:, 1::2], fe)
    pe, pe2 = ckpt["vision_encoder.visual.pos_embed"], ck2["vision_encoder.visual.pos_embed"]
    assert pe2.shape == (1, 1 + 28 * 28, pe.shape[-1]) and torch.equal(pe2[0, 0], pe[0, 0])
    g = pe[0, 1:].reshape(14, 14, -1)
    g2 = pe2[0, 1:].reshape(28, 28, -1)
    assert torch.allclose(g2[1::2, 1::2][:-1, :-1], (0.75 * 0.75 * g[:-1, :-1] + 0.75 * 0.25 * (g[1:, :-1] + g[:-1, 1:]) + 0.0625 * g[1:, 1:]), atol=1e-5)
    txt, _ = demo.load_from_pretrained_dir(pdir, return_modal="text")
    assert txt and all(k.startswith("bert.") or k.startswith("cls.") for k in txt)

    from PIL import Image
    img = (np.random.RandomState(1).rand(50, 80, 3) * 255).astype(np.uint8)
    f = str(tmp_path / "a.png")
    Image.fromarray(img).save(f)
    p = ImageProcessor(224, "swin")
    assert p.mean == [0.485, 0.456, 0.406] and ImageProcessor(224, "evaclip01_giant").mean[0] == 0.48145466
    x = p(f)
    assert x.shape == (1, 3, 224, 224)
    p50 = ImageProcessor(50, "swin")
    p50.resolution = 50
    same = p50.transform(torch.from_numpy(img[:, :50].copy()).permute(2, 0, 1).float() / 255)   # identity resize
    ref = (torch.from_numpy(img[:, :50].copy()).permute(2, 0, 1).float() / 255 - torch.tensor(p.mean).view(3, 1, 1)) / torch.tensor(p.std).view(3, 1, 1)
    assert torch.allclose(same, ref, atol=1e-6)
    assert p(str(tmp_path / "missing.jpg")) is None
    open(str(tmp_path / "bad.jpg"), "w").write("x")
    assert p(str(tmp_path / "bad.jpg")) is None


def test_loader_against_reference_golden(tmp_path):
    """tests/golden/ckpt_remap.pt holds outputs of the reference's own load_from_pretrained_dir (oracle/make_golden.py ckpt):
    same keys, dtypes and values, bit-exact, for the EVA and OpenAI-CLIP table layouts and every return_modal."""
    import inference_demo as demo
    cases = torch.load(os.path.join(os.path.dirname(__file__), "golden", "ckpt_remap.pt"))
    for name, c in cases.items():
        d = str(tmp_path / name)
        os.makedirs(os.path.join(d, "ckpt"))
        os.makedirs(os.path.join(d, "log"))
        json.dump(c["hps"], open(os.path.join(d, "log", "hps.json"), "w"))
        torch.save({"stale": torch.zeros(1)}, os.path.join(d, "ckpt", "model_step_9.pt"))
        torch.save(c["stored"], os.path.join(d, "ckpt", "model_step_10.pt"))
        for modal, want in c["outs"].items():
            got, cfg = demo.load_from_pretrained_dir(d, return_modal=modal)
            assert set(got) == set(want), (name, modal)
            for k in want:
                assert got[k].dtype == want[k].dtype and got[k].shape == want[k].shape and torch.equal(got[k], want[k]), (name, modal, k)
            assert cfg.max_vision_sample_num == 8 and cfg["vision_resolution"] == 10
