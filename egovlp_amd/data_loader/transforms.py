"""Device-side counterpart of the reference's TRAIN transform (data_loader/transforms.py:14-19):

    RandomResizedCrop(input_res, scale=randcrop_scale) -> RandomHorizontalFlip() -> ColorJitter(0, 0, 0) -> Normalize(mean, std)

The reference runs it on the host, on float frames, and ships 4 x 3 x 224 x 224 floats per clip to the GPU.  Here only the
RANDOM DRAWS stay on the host -- `train_transform_params` returns, per clip, the crop box and the flip flag -- and the pixels
are produced by `egv_patch_gather_u8_aug` straight from the decoded uint8 clip inside the patch gather of the video encoder
(`SpaceTimeTransformer.set_input_augmentation`).  One box per clip, as in the reference (the transform is applied to the
[T, C, H, W] tensor as a whole).  The 'val' / 'test' transforms are deterministic resizes of the loader and are not rebuilt.

The box sampling restates torchvision 0.13's `RandomResizedCrop.get_params` (third-party, pinned in the reference's
environment.yml): ten tries of (area fraction ~ U(scale), log-aspect ~ U(log ratio)), then the central fallback crop.
"""
import math

import torch


def random_resized_crop_box(height, width, scale=(0.5, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0), generator=None):
    area = height * width
    log_ratio = (math.log(ratio[0]), math.log(ratio[1]))
    for _ in range(10):
        target_area = area * float(torch.empty(1).uniform_(scale[0], scale[1], generator=generator))
        aspect = math.exp(float(torch.empty(1).uniform_(log_ratio[0], log_ratio[1], generator=generator)))
        w = int(round(math.sqrt(target_area * aspect)))
        h = int(round(math.sqrt(target_area / aspect)))
        if 0 < w <= width and 0 < h <= height:
            i = int(torch.randint(0, height - h + 1, (1,), generator=generator))
            j = int(torch.randint(0, width - w + 1, (1,), generator=generator))
            return i, j, h, w
    in_ratio = float(width) / float(height)
    if in_ratio < min(ratio):
        w = width
        h = int(round(w / min(ratio)))
    elif in_ratio > max(ratio):
        h = height
        w = int(round(h * max(ratio)))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w


def train_transform_params(batch, height, width, randcrop_scale=(0.5, 1.0), flip_p=0.5, generator=None):
    """-> int32 [batch, 5] (top, left, h, w, flip): the host half of the fused train transform."""
    rows = []
    for _ in range(batch):
        i, j, h, w = random_resized_crop_box(height, width, randcrop_scale, generator=generator)
        flip = int(float(torch.rand(1, generator=generator)) < flip_p)
        rows.append([i, j, h, w, flip])
    return torch.tensor(rows, dtype=torch.int32)
