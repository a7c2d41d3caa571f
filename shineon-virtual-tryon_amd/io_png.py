"""Inter-stage wire format: the warp stage hands its output to the try-on stage as PNG files
(reference: visualization.py:56-88; warp_model.py:119-149; unet_mask_model.py:258-280).
Quantisation rule: ((t + 1) * 0.5 * 255) clamped to [0, 255], TRUNCATED to uint8, CHW -> HWC."""
import os

import numpy as np
import torch


def get_save_paths(save_dirs, img_names):
    return [os.path.join(s, i) for s, i in zip(save_dirs, img_names)]


def tensor_to_uint8(img_tensor):
    """(C, H, W) float in [-1, 1] -> uint8 array (H, W) or (H, W, 3), bit-exact with the reference."""
    t = (img_tensor.detach().to(torch.float32).cpu().contiguous().clone() + 1) * 0.5 * 255
    arr = t.clamp(0, 255).numpy().astype("uint8")
    if arr.shape[0] == 1:
        return arr.squeeze(0)
    if arr.shape[0] == 3:
        return arr.swapaxes(0, 1).swapaxes(1, 2)
    raise ValueError(f"Trying to save an image that is not 1 or 3 channels; this is unexpected. {arr.shape=}")


def save_images(img_tensors, img_names, save_dirs):
    from PIL import Image

    if len(save_dirs) == 1:
        save_dirs = [save_dirs] * len(img_names)
    for img_tensor, img_name, save_dir in zip(img_tensors, img_names, save_dirs):
        if "warp-mask" in save_dir and "VitonDataset" not in save_dir:
            continue
        path = os.path.join(save_dir, img_name)
        if os.path.exists(path):
            continue
        os.makedirs(os.path.dirname(path), exist_ok=True)
        Image.fromarray(tensor_to_uint8(img_tensor)).save(path)
