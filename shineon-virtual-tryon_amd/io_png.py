"""Inter-stage wire format: the warp stage hands its output to the try-on stage as image files
(reference: visualization.py:56-88 writer; warp_model.py:119-149 / unet_mask_model.py:258-280 call sites;
datasets/vvt_dataset.py:139-150 + tryon_dataset.py:109-118,148-152 reader).

Write: byte = trunc(clamp((t + 1) * 0.5 * 255, 0, 255)), CHW -> HWC; done for the whole batch in one kernel on the GPU
(dataprep.quantise_images), so only a quarter of the bytes cross PCIe.  Read: (byte / 255 - 0.5) / 0.5, again one
kernel per batch (dataprep.images_to_normed).  Files are encoded / decoded by PIL like in the reference, so the
extension of the cloth / image name decides the container (.png lossless, .jpg lossy).
"""
import os

import numpy as np
import torch


def get_save_paths(save_dirs, img_names):
    return [os.path.join(s, i) for s, i in zip(save_dirs, img_names)]


def tensor_to_uint8(img_tensor):
    """One (C, H, W) image -> uint8 (H, W) or (H, W, 3); CPU tensors use torch arithmetic, GPU tensors the HIP kernel."""
    if img_tensor.shape[0] not in (1, 3):
        raise ValueError("Trying to save an image that is not 1 or 3 channels; this is unexpected. "
                         f"array.shape={tuple(img_tensor.shape)}")
    return _batch_to_uint8(img_tensor.unsqueeze(0))[0]


def _batch_to_uint8(batch):
    """(N, C, H, W) float -> list of N uint8 arrays, HWC (C == 3) or HW (C == 1)."""
    t = batch.detach().to(torch.float32)
    c = t.shape[1]
    if c not in (1, 3):
        raise ValueError(f"Trying to save an image that is not 1 or 3 channels; this is unexpected. array.shape={tuple(t.shape[1:])}")
    if t.is_cuda:
        from .dataprep import quantise_images

        q = quantise_images(t).cpu().numpy()
    else:
        q = ((t + 1) * 0.5 * 255).clamp(0, 255).numpy().astype(np.uint8).transpose(0, 2, 3, 1)
    return [a[..., 0] if c == 1 else a for a in q]


def _wanted(save_dir, path):
    """The reference's two skip rules: warp masks are only kept for VitonDataset, existing files are never rewritten."""
    if "warp-mask" in save_dir and "VitonDataset" not in save_dir:
        return False
    return not os.path.exists(path)


def save_images(img_tensors, img_names, save_dirs):
    """Save a batch of image tensors under save_dirs[i] / img_names[i] (one shared dir is broadcast)."""
    from PIL import Image

    dirs = list(save_dirs) * len(img_names) if len(save_dirs) == 1 else list(save_dirs)
    todo = [(i, os.path.join(d, n), d) for i, (n, d) in enumerate(zip(img_names, dirs))]
    todo = [(i, path) for i, path, d in todo if _wanted(d, path)]
    if not todo:
        return []
    batch = img_tensors if isinstance(img_tensors, torch.Tensor) else torch.stack(list(img_tensors))
    arrays = _batch_to_uint8(batch[[i for i, _ in todo]] if len(todo) != batch.shape[0] else batch)
    for (_, path), array in zip(todo, arrays):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        Image.fromarray(array).save(path)
    return [path for _, path in todo]


class StageWriter:
    """The skip-or-run-and-write pattern both `test_step`s share (warp_model.py:115-152, unet_mask_model.py:250-282):
    the batch is skipped when every file of its PRIMARY output already exists; otherwise `produce()` runs the model
    and returns {subdir: (N, C, H, W) tensor} whose images are written as <root>/<dataset_name[i]>/<subdir>/<name[i]>."""

    def __init__(self, root, dataset_names, names, primary):
        self.root, self.dataset_names, self.names, self.primary = root, list(dataset_names), list(names), primary

    def dirs(self, subdir):
        return [os.path.join(self.root, d, subdir) for d in self.dataset_names]

    def done(self):
        return all(os.path.exists(p) for p in get_save_paths(self.dirs(self.primary), self.names))

    def run(self, produce):
        if self.done():
            return {"progress_bar": {"file": f"Skipping {self.names[0]}"}}
        for subdir, tensor in produce().items():
            save_images(tensor, self.names, self.dirs(subdir))
        return {"progress_bar": {"file": f"{self.names[0]}"}}


# ---- reader side (try-on stage input) ------------------------------------------------------------------------------
def read_image_u8(path, channels=3):
    """Decode one image file to a uint8 (H, W, channels) array, the way Image.open(...) feeds ToTensor."""
    from PIL import Image

    img = Image.open(path)
    if channels == 3 and img.mode != "RGB":
        img = img.convert("RGB")
    elif channels == 1 and img.mode != "L":
        img = img.convert("L")
    a = np.array(img, copy=True)
    return a if a.ndim == 3 else a[:, :, None]


def load_images(paths, device, channels=3):
    """Files -> (N, channels, H, W) fp32 in [-1, 1] = ToTensor + Normalize(0.5, 0.5) (tryon_dataset.py:109-118,148-152):
    decoded bytes are uploaded as uint8 and dequantised by one kernel on the GPU."""
    from .dataprep import images_to_normed

    u8 = torch.from_numpy(np.stack([read_image_u8(p, channels) for p in paths]))
    return images_to_normed(u8.to(device, non_blocking=True))


def find_warp_cloth(warp_cloth_dir, cloth_name, dataset_name=None):
    """Where the warp stage's test_step left the warped cloth for `cloth_name`: <dir>/<cloth_name>, or the
    <dir>/<Dataset>/warp-cloth/<cloth_name> layout directly under a warp run's result folder."""
    cands = [os.path.join(warp_cloth_dir, cloth_name)]
    if dataset_name:
        cands.append(os.path.join(warp_cloth_dir, dataset_name, "warp-cloth", cloth_name))
    for c in cands:
        if os.path.exists(c):
            return c
    raise FileNotFoundError(f"no warped cloth for {cloth_name} under {warp_cloth_dir} (tried {cands}); "
                            "run the warp stage's test first or pass --warp_cloth_dir")
