"""Image export of the evaluation path (models/rcf_model.py:241-273): a stand-in for
`torchvision.utils.save_image` (make_grid + round-to-u8 + PIL encode), which the reference uses to write the
evaluation JPEG and the `pred_seg_*.png` masks that the offline stages (CRF post-processing, semantic
constraints) read back.  Host-side plumbing: nothing here is on the training hot path.
"""
import math
import os

import torch


def make_grid(t, nrow=8, padding=2, pad_value=0.0):
    """torchvision.utils.make_grid for a [B,C,H,W] (or [C,H,W] / [H,W]) float tensor, default arguments."""
    if t.dim() == 2:
        t = t.unsqueeze(0)
    if t.dim() == 3:
        if t.shape[0] == 1:
            t = torch.cat((t, t, t), 0)
        t = t.unsqueeze(0)
    if t.dim() == 4 and t.shape[1] == 1:
        t = torch.cat((t, t, t), 1)
    if t.shape[0] == 1:
        return t.squeeze(0)
    nmaps = t.shape[0]
    xmaps = min(nrow, nmaps)
    ymaps = int(math.ceil(float(nmaps) / xmaps))
    height, width = int(t.shape[2] + padding), int(t.shape[3] + padding)
    grid = t.new_full((t.shape[1], height * ymaps + padding, width * xmaps + padding), pad_value)
    k = 0
    for y in range(ymaps):
        for x in range(xmaps):
            if k >= nmaps:
                break
            grid.narrow(1, y * height + padding, height - padding).narrow(2, x * width + padding, width - padding).copy_(t[k])
            k += 1
    return grid


def to_uint8_hwc(t):
    """the rounding of torchvision.utils.save_image: x*255 + 0.5, clamp, truncate"""
    grid = make_grid(t)
    return grid.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to("cpu", torch.uint8).numpy()


def save_image(t, path):
    from PIL import Image                               # present in the image; fails loudly if not
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    Image.fromarray(to_uint8_hwc(t.detach().float())).save(path)
