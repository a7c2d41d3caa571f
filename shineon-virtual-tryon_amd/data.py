"""Synthetic VVT-shaped batches and the dataset registry hook.

The reference's file-backed datasets (datasets/*.py) are out of scope; when its `datasets` package is
importable it is used as is, otherwise `--dataset synthetic` provides tensors with the same keys, shapes
and value ranges as datasets/tryon_dataset.py:481-537 builds.
"""
import numpy as np
import torch
from torch.utils.data import Dataset

from . import tryon_channels as ch

_KEYS = {
    "image": ch.RGB_CHANNELS, "prev_image": ch.RGB_CHANNELS, "cloth": ch.CLOTH_CHANNELS,
    "cloth_mask": ch.CLOTH_MASK_CHANNELS, "im_cloth": ch.RGB_CHANNELS, "grid_vis": ch.RGB_CHANNELS,
    "agnostic": ch.AGNOSTIC_CHANNELS, "cocopose": ch.COCOPOSE_CHANNELS, "densepose": ch.DENSEPOSE_CHANNELS,
    "flow": ch.FLOW_CHANNELS, "silhouette": ch.SILHOUETTE_CHANNELS, "im_head": ch.RGB_CHANNELS,
    "im_cocopose": 1,
}
_MASKS = ("cloth_mask", "silhouette")


def _smooth_field(rng, shape):
    """Band-limited image in [-1, 1]: two random low-frequency sinusoids per channel.  Used by the parity
    fixtures: with white-noise images d(grid_sample)/d(grid) flips with every tap change, which turns a 1e-5
    difference in the grid into O(1) differences in the gradients (an input property, not a kernel one)."""
    n, c, h, w = shape
    y = np.arange(h, dtype=np.float64)[:, None] / h
    x = np.arange(w, dtype=np.float64)[None, :] / w
    out = np.empty(shape, np.float32)
    for f in range(n):
        for k in range(c):
            fx, fy = rng.integers(1, 4, 2), rng.integers(1, 4, 2)
            ph = rng.uniform(0, 2 * np.pi, 2)
            out[f, k] = (0.55 * np.sin(2 * np.pi * (fx[0] * x + fy[0] * y) + ph[0])
                         + 0.4 * np.sin(2 * np.pi * (fx[1] * x - fy[1] * y) + ph[1]))
    return out


def synthetic_sample(index, height=256, width=192, n_frames=1, seed=420, radius=5, smooth=False):
    """One sample of the batch dict: images U(-1,1) (or band-limited when smooth=True), masks Bernoulli(.5),
    cocopose -1 with 11x11 +1 squares, flow N(0, 2) px; tensors are (C, H, W) or (n_frames, C, H, W)."""
    rng = np.random.default_rng([seed, index])
    out = {}
    for key, c in _KEYS.items():
        shape = (n_frames, c, height, width)
        if key in _MASKS:
            a = (rng.random(shape) < 0.5).astype(np.float32)
        elif key == "cocopose":
            a = -np.ones(shape, np.float32)
            for f in range(n_frames):
                for k in range(c):
                    y, x = int(rng.integers(0, height)), int(rng.integers(0, width))
                    a[f, k, max(0, y - radius):y + radius + 1, max(0, x - radius):x + radius + 1] = 1.0
        elif key == "flow":
            a = rng.normal(0.0, 2.0, shape).astype(np.float32)
        elif smooth:
            a = _smooth_field(rng, shape)
        else:
            a = rng.uniform(-1.0, 1.0, shape).astype(np.float32)
        t = torch.from_numpy(a)
        out[key] = t if n_frames > 1 else t[0]
    name = f"synthetic_{index:06d}.png"
    for key in ("dataset_name", "cloth_name", "cloth_path", "image_name", "image_path"):
        val = "SyntheticDataset" if key == "dataset_name" else name
        # like the reference's NFramesInterface (n_frames_interface.py:79-100): strings arrive as one list per
        # frame, which maybe_combine_frames_and_channels unwraps when n_frames_total == 1
        out[key] = [val] * n_frames
    return out


class SyntheticDataset(Dataset):
    device_resident = True  # trainer.Trainer keeps the collated set in HBM and gathers batches on the device

    def __init__(self, opt, length=None):
        self.opt = opt
        self.length = length if length is not None else getattr(opt, "synthetic_length", 64)

    def __len__(self):
        return self.length

    def __getitem__(self, index):
        o = self.opt
        return synthetic_sample(index, o.fine_height, o.fine_width, getattr(o, "n_frames_total", 1))

    def make_validation_dataset(self, opt):
        return SyntheticDataset(opt, length=max(1, self.length // 8))

    @staticmethod
    def modify_commandline_options(parser, is_train):
        parser.add_argument("--fine_width", type=int, default=192)
        parser.add_argument("--fine_height", type=int, default=256)
        parser.add_argument("--radius", type=int, default=5)
        parser.add_argument("--synthetic_length", type=int, default=64)
        parser.add_argument("--n_frames_total", type=int, default=1, metavar="N")
        parser.add_argument("--n_frames_now", type=int, default=None, metavar="N")
        parser.add_argument("--visualize_flow", action="store_true")
        return parser


def synthetic_batch(batch_size, device, height=256, width=192, n_frames=1, seed=420, start=0, smooth=False):
    """Collated batch of synthetic samples already resident on `device`."""
    from torch.utils.data.dataloader import default_collate

    batch = default_collate([synthetic_sample(start + i, height, width, n_frames, seed, smooth=smooth)
                             for i in range(batch_size)])
    return {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}


class _FileBackedDataset(Dataset):
    """Command-line surface of the reference's file-backed datasets (datasets/tryon_dataset.py:63-96,
    n_frames_interface.py:33-52, vvt_dataset.py:22-31, viton_dataset.py:14-18, mpv_dataset.py:15-18) so that its documented
    command lines (docs/2_inference.md, docs/3_train.md) parse and resolve here.  Reading those folder layouts is outside
    the hot path (the boundary is the batch dict): with the reference's `datasets` package on sys.path it is used
    as is, otherwise constructing one of these raises."""

    extra_flags = ()

    def __init__(self, opt, *a, **k):
        raise NotImplementedError(
            f"--dataset {getattr(opt, 'dataset', '?')} reads the reference's folder layout: put the reference's datasets/ "
            "package on sys.path (it is used unchanged - it produces the batch dict this package consumes), or use "
            "--dataset synthetic")

    @classmethod
    def modify_commandline_options(cls, parser, is_train):
        parser.add_argument("--val_fraction", type=float, default=0.01 if is_train else 0)
        parser.add_argument("--cloth_mask_threshold", type=int, default=240)
        parser.add_argument("--image_scale", type=float, default=1)
        parser.add_argument("--fine_width", type=int, default=192)
        parser.add_argument("--fine_height", type=int, default=256)
        parser.add_argument("--radius", type=int, default=5)
        parser.add_argument("--visualize_flow", action="store_true")
        parser.add_argument("--n_frames_total", type=int, default=1, metavar="N")
        parser.add_argument("--n_frames_now", type=int, default=None, metavar="N")
        for flag, kw in cls.extra_flags:
            parser.add_argument(flag, **kw)
        return parser


class _VVT(_FileBackedDataset):
    extra_flags = (("--vvt_dataroot", dict(default="/data_hdd/fw_gan_vvt")), ("--warp_cloth_dir", dict(default=None)))


class _Viton(_FileBackedDataset):
    extra_flags = (("--viton_dataroot", dict(default="data")), ("--data_list", dict(default="train_pairs.txt")))


class _MPV(_FileBackedDataset):
    extra_flags = (("--mpv_dataroot", dict(default="/data_hdd/mpv_competition")),)


class _VitonVvtMpv(_FileBackedDataset):
    extra_flags = _Viton.extra_flags + _VVT.extra_flags + _MPV.extra_flags


_FILE_BACKED = {"vvt": _VVT, "viton": _Viton, "mpv": _MPV, "viton_vvt_mpv": _VitonVvtMpv}


def find_dataset_using_name(name):
    if name == "synthetic":
        return SyntheticDataset
    try:  # the reference's own datasets package, if it is on sys.path
        import datasets as ref_datasets

        return ref_datasets.find_dataset_using_name(name)
    except Exception as e:  # noqa: BLE001
        if name in _FILE_BACKED:   # flags parse; construction explains what is needed
            return _FILE_BACKED[name]
        raise NotImplementedError(
            f"dataset '{name}' needs the reference's datasets/ package on sys.path ({type(e).__name__}: {e}); "
            "use --dataset synthetic otherwise"
        ) from e


def get_option_setter(name):
    return find_dataset_using_name(name).modify_commandline_options
