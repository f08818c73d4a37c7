"""Host-side image preprocessing with the reference's surface (model/imageprocessor.py:10-63): PIL decode -> RGB -> [0,1] CHW
float tensor -> Resize((r, r)) -> Normalize(mean, std) -> (1, 3, r, r).  Mean/std are chosen by the encoder-type string exactly
as the reference does (CLIP statistics for 'clip*'/'evaclip*', ImageNet otherwise - note inference_demo.py passes "swin").
torchvision is not available here: Resize is restated as bilinear interpolation without antialiasing, which is what
torchvision 0.15's Resize does for tensor inputs (resize parity itself is unpinned).  Decoding/resizing is
host I/O outside the hot path; the tensor it emits is where the MI355X path starts."""
import os

import numpy as np
import torch
import torch.nn.functional as F


def image_stats(encoder_type):
    """imageprocessor.py:17-22 / videoprocessor.py:27-33: CLIP statistics for clip* / evaclip*, ImageNet otherwise."""
    if encoder_type.startswith("clip") or encoder_type.startswith("evaclip"):
        return [0.48145466, 0.4578275, 0.40821073], [0.26862954, 0.26130258, 0.27577711]
    return [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]


class ImageProcessor(object):
    def __init__(self, image_resolution, image_encoder_type, image_transforms="none", training=True, device=None):
        """device=None: host path (torch ops); device="cuda": decode on the host, then ToTensor + Resize + Normalize in one
        device kernel (mico_image_preprocess) - the returned tensor already lives on the device."""
        self.training = training
        self.resolution = image_resolution
        self.image_encoder_type = image_encoder_type
        self.device = device
        self.mean, self.std = image_stats(image_encoder_type)
        if image_transforms != "none":
            raise NotImplementedError(image_transforms)
        self.image_transforms = image_transforms

    def transform(self, img):
        """img: float CHW tensor in [0,1] -> resized + normalised CHW."""
        r = self.resolution
        img = F.interpolate(img.unsqueeze(0), size=(r, r), mode="bilinear", align_corners=False, antialias=False)[0]
        mean = torch.tensor(self.mean, dtype=img.dtype).view(3, 1, 1)
        std = torch.tensor(self.std, dtype=img.dtype).view(3, 1, 1)
        return (img - mean) / std

    def __call__(self, image_file):
        try:
            if not os.path.exists(image_file):
                print("not have image", image_file)
                return None
            from PIL import Image
            img = Image.open(image_file).convert("RGB")
            u8 = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy())
            if self.device is not None:
                from .videoprocessor import preprocess_frames_device
                return preprocess_frames_device(u8.unsqueeze(0), self.resolution, self.mean, self.std, self.device)
            return self.transform(u8.permute(2, 0, 1).float().div(255.0)).unsqueeze(0)
        except Exception as e:   # the reference swallows errors and returns None (imageprocessor.py:61-63)
            print(e)
            return None
