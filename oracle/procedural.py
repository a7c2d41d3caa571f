"""ORACLE SUPPORT — TEST INFRASTRUCTURE ONLY.

Seed-addressed procedural weights (pure numpy, float64 -> fp32): the golden-vector generator (which loads
them into the REFERENCE's modules), the oracle and the HIP modules all regenerate identical parameters
from (seed, state_dict key), so no 76-91 MB state_dict has to be stored.
"""
import zlib

import numpy as np
import torch


def _rng(seed, key):
    return np.random.default_rng([int(seed), zlib.crc32(key.encode())])


def procedural_tensor(key, shape, seed=420):
    """Distribution chosen from the key name so that activations stay O(1) through the stacks."""
    shape = tuple(int(s) for s in shape)
    r = _rng(seed, key)
    leaf = key.split(".")[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros((), dtype=torch.long)
    if leaf == "running_mean":
        return torch.from_numpy(r.normal(0.0, 0.1, shape)).float()
    if leaf == "running_var":
        return torch.from_numpy(r.uniform(0.5, 1.5, shape)).float()
    if leaf == "gamma":  # SelfAttention: the reference initialises 0, which would hide the attention path
        return torch.full(shape, 0.7, dtype=torch.float32)
    if leaf == "bias":
        return torch.from_numpy(r.normal(0.0, 0.05, shape)).float()
    if leaf in ("weight_u", "weight_v"):  # spectral norm's power-iteration vectors: unit length, NOT converged
        x = r.normal(0.0, 1.0, shape)
        return torch.from_numpy(x / np.linalg.norm(x)).float()
    if leaf == "weight_orig":
        leaf = "weight"
    if leaf == "weight" and len(shape) == 1:  # BatchNorm scale
        return torch.from_numpy(r.normal(1.0, 0.1, shape)).float()
    if leaf == "weight":
        fan_in = int(np.prod(shape[1:]))
        std = (2.0 / fan_in) ** 0.5  # He: keeps 13 VGG layers / 12 U-Net layers in range
        if "linear" in key or "query_conv" in key or "key_conv" in key:
            std = 0.02  # the reference's own init scale; keeps the attention softmax soft (well conditioned)
        if "mlp_gamma" in key or "mlp_beta" in key:
            std *= 0.25  # SPADE modulation: up to 4 x (1 + gamma) factors per norm site, 2-3 sites per block
        return torch.from_numpy(r.normal(0.0, std, shape)).float()
    raise KeyError(f"no procedural rule for {key}")


def procedural_state_dict(shapes, seed=420):
    """shapes: {key: shape} (e.g. from module.state_dict()) -> {key: tensor}."""
    return {k: procedural_tensor(k, s, seed) for k, s in shapes.items()}


def shapes_of(state_dict):
    return {k: tuple(v.shape) for k, v in state_dict.items()}
