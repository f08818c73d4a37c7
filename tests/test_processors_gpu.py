"""Device-side preprocessing kernels (f3).  Audio: normalise / zero-pad / window selection after the filterbank against the
reference's own post-filterbank code (tests/golden/processors.pt), bit-exact.  Image / video frames: ToTensor + bilinear Resize +
Normalize in one kernel against the host path (torch bilinear without antialias = torchvision 0.15's tensor Resize; the resize
arithmetic itself is unpinned third-party code)."""
import os

import numpy as np
import pytest
import torch

from common import golden

pytestmark = pytest.mark.gpu


def test_fbank_windows_match_reference(cuda):
    from mico_amd.model.audioprocessor import AudioProcessor
    for case in golden("processors.pt")["audio"]:
        proc = AudioProcessor(case["melbins"], case["target_length"], case["sample_num"], resize_melbin_num=case["melbins"],
                              training=False, device=cuda)
        got = proc.from_fbank(case["fbank"])
        assert got.shape == case["out"].shape
        assert (got.cpu() - case["out"]).abs().max() <= 1e-6 * case["out"].abs().max()


def test_image_and_frames(cuda, tmp_path):
    from PIL import Image
    from mico_amd.model.imageprocessor import ImageProcessor
    from mico_amd.model.videoprocessor import VideoProcessor
    rng = np.random.RandomState(3)
    for (h, w) in ((428, 640), (224, 224), (97, 301), (600, 180)):
        img = (rng.rand(h, w, 3) * 255).astype(np.uint8)
        f = str(tmp_path / f"i_{h}_{w}.png")
        Image.fromarray(img).save(f)
        for enc in ("swin", "evaclip01_giant"):
            host = ImageProcessor(224, enc)(f)
            dev = ImageProcessor(224, enc, device=cuda)(f)
            assert dev.is_cuda and dev.shape == (1, 3, 224, 224)
            assert (dev.cpu() - host).abs().max() < 2e-6 * host.abs().max().clamp_min(1.0), (h, w, enc)
    d = tmp_path / "frames"
    os.makedirs(d)
    frames = []
    for i in range(10):
        a = (rng.rand(120, 160, 3) * 255).astype(np.uint8)
        Image.fromarray(a).save(str(d / f"img_{i + 1:04d}.png"))
        frames.append(a)
    vp = VideoProcessor(224, "evaclip01_giant", sample_num=4, training=False, device=cuda)
    out = vp(str(d))
    assert out.shape == (4, 3, 224, 224)
    picks = [1, 4, 6, 8]           # split(range(10), 4) = [0-2][3-5][6-7][8-9] -> element (len + 1) // 2 - 1 of each
    ip = ImageProcessor(224, "evaclip01_giant")
    for k, i in enumerate(picks):
        ref = ip.transform(torch.from_numpy(frames[i]).permute(2, 0, 1).float() / 255)
        assert (out[k].cpu() - ref).abs().max() < 2e-6 * ref.abs().max()
    assert vp(str(tmp_path / "missing")) is None
