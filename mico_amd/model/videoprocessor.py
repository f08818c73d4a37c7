"""Video / frame-folder preprocessing with the reference's surface (model/videoprocessor.py:11-108): `split` the frame list into
sample_num contiguous groups (padding with the last frame), pick one frame per group (random in training, the middle one in
evaluation), decode (PIL, host), then ToTensor + Resize + Normalize ON THE DEVICE in one kernel (mico_image_preprocess).  The
'raw' container format needs decord, which this image does not have; frame folders ('frame') are supported."""
import os
import random

import numpy as np
import torch

from .. import _lib
from .imageprocessor import image_stats


def split(frame_name_lists, sample_num):
    """videoprocessor.py:11-15 / audioprocessor.py:8-12: sample_num contiguous groups whose sizes differ by at most one (the
    first len % sample_num groups get the extra element); a short list is first padded with its last element."""
    items = list(frame_name_lists)
    while len(items) < sample_num:
        items.append(items[-1])
    base, extra = divmod(len(items), sample_num)
    groups, start = [], 0
    for g in range(sample_num):
        size = base + (1 if g < extra else 0)
        groups.append(items[start:start + size])
        start += size
    return groups


def sample_indices(groups, training):
    """one element per group: random.choice in training, the (upper-)middle element otherwise (videoprocessor.py:66-69)."""
    if training:
        return [random.choice(i) for i in groups]
    return [i[(len(i) + 1) // 2 - 1] for i in groups]


def preprocess_frames_device(frames_u8, resolution, mean, std, device="cuda"):
    """uint8 [n, H, W, 3] (host or device) -> normalised fp32 [n, 3, r, r] on the device; see mico_image_preprocess."""
    x = frames_u8.to(device).contiguous()
    assert x.dtype == torch.uint8 and x.dim() == 4 and x.shape[-1] == 3
    n, H, W, _ = x.shape
    out = torch.empty((n, 3, resolution, resolution), dtype=torch.float32, device=x.device)
    rc = _lib.lib().mico_image_preprocess(x.data_ptr(), n, H, W, out.data_ptr(), resolution, resolution, mean[0], mean[1], mean[2],
                                          1.0 / std[0], 1.0 / std[1], 1.0 / std[2], torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(rc, "mico_image_preprocess")
    return out


class VideoProcessor(object):
    def __init__(self, video_resolution, video_encoder_type, sample_num=4, video_transforms="none", data_format="frame", training=True,
                 device="cuda"):
        self.frame_syncaug = True
        self.training = training
        self.sample_num = sample_num
        self.data_format = data_format
        self.resolution = video_resolution
        self.video_encoder_type = video_encoder_type
        self.mean, self.std = image_stats(video_encoder_type)
        self.device = device
        if video_transforms != "none":
            raise NotImplementedError(video_transforms)

    def __call__(self, video_file):
        try:
            if self.data_format != "frame":
                raise NotImplementedError("data_format='raw' decodes with decord, which is not available; extract frames to a folder")
            if not os.path.exists(video_file):
                print("not have videos", video_file)
                return None
            from PIL import Image
            frames = sorted(os.listdir(video_file))
            picked = sample_indices(split(frames, self.sample_num), self.training)
            imgs = [np.asarray(Image.open(os.path.join(video_file, f)).convert("RGB"), dtype=np.uint8) for f in picked]
            batch = torch.from_numpy(np.stack(imgs, 0))
            return preprocess_frames_device(batch, self.resolution, self.mean, self.std, self.device)
        except Exception as e:   # the reference swallows errors and returns None (videoprocessor.py:104-107)
            print(e)
            print(video_file)
            return None
