"""Batch plumbing that runs first in every step (reference: datasets/n_frames_interface.py:105-138,
util/__init__.py:64-66)."""
import collections.abc

import torch

from . import ops


def maybe_combine_frames_and_channels(opt, inputs, has_batch_dim=True):
    """(b, n, c, h, w) -> (b, n*c, h, w) views; unwrap singleton string lists when n_frames_total == 1."""
    if not hasattr(opt, "n_frames_total"):
        return inputs

    def combine(t):
        if isinstance(t, torch.Tensor):
            if has_batch_dim and t.dim() == 5:
                bs, n, c, h, w = t.shape
                t = t.reshape(bs, n * c, h, w)
            elif not has_batch_dim and t.dim() == 4:
                n, c, h, w = t.shape
                t = t.reshape(n * c, h, w)
        elif isinstance(t, collections.abc.Sequence) and not isinstance(t, str):
            if opt.n_frames_total == 1:
                t = t[0]
        return t

    return {k: combine(v) for k, v in inputs.items()}


def get_and_cat_inputs(batch, names):
    """Channel concatenation of the named batch tensors, in the given (sorted) order, as one NHWC slab."""
    tensors = [batch[n] for n in names]
    if len(tensors) == 1:
        return tensors[0]
    if all(t.is_cuda for t in tensors):
        return ops.cat_channels(tensors)
    return torch.cat(tensors, dim=1)
