"""Host half of the fused train transform: the crop-box sampler restates torchvision's RandomResizedCrop.get_params."""
import math

import torch

from egovlp_amd.data_loader.transforms import random_resized_crop_box, train_transform_params


def test_boxes_lie_inside_the_frame_and_respect_scale_and_ratio():
    g = torch.Generator().manual_seed(0)
    H, W = 256, 341
    fr = []
    for _ in range(400):
        i, j, h, w = random_resized_crop_box(H, W, (0.5, 1.0), generator=g)
        assert 0 <= i and i + h <= H and 0 <= j and j + w <= W and h > 0 and w > 0
        fr.append(h * w / (H * W))
        assert 3 / 4 / 1.05 <= w / h <= 4 / 3 * 1.05 or (h, w) == (H, W)
    assert 0.49 <= min(fr) and max(fr) <= 1.0 and 0.65 < sum(fr) / len(fr) < 0.85


def test_params_tensor_layout_and_flip_rate():
    g = torch.Generator().manual_seed(1)
    p = train_transform_params(512, 224, 224, generator=g)
    assert p.dtype == torch.int32 and tuple(p.shape) == (512, 5)
    assert set(p[:, 4].tolist()) <= {0, 1} and 0.4 < float(p[:, 4].float().mean()) < 0.6
    assert bool(((p[:, 0] + p[:, 2]) <= 224).all()) and bool(((p[:, 1] + p[:, 3]) <= 224).all())


def test_fallback_is_the_central_crop_with_clamped_ratio():
    # a 10:1 frame never admits a 3/4..4/3 box of >= 99 % of its area: the sampler falls back to the central crop
    i, j, h, w = random_resized_crop_box(100, 1000, scale=(0.99, 1.0), generator=torch.Generator().manual_seed(2))
    assert (h, w) == (100, int(round(100 * 4 / 3))) and i == 0 and j == (1000 - w) // 2
    assert math.isclose(w / h, 4 / 3, rel_tol=0.02)
