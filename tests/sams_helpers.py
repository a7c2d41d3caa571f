"""Shared by the SAMS-GAN tests: the option sets of the committed goldens (tests/golden/make_golden.py::SAMS_VARIANTS),
and the three training steps run through the oracle in Lightning's order."""
import argparse
import contextlib
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))

SAMS_VARIANTS = {
    "base": dict(),
    "attn_gelu": dict(attention_middle_indices=["0"], attention_decoder_indices=["-1"], activation="gelu", gan_mode="ls",
                      norm_G="spectralspadebatch3x3"),
    "progressive": dict(n_frames_total=4, n_frames_now=2, flow_warp=False, gan_mode="original", norm_G="spadeinstance3x3",
                        norm_D="spectralbatch", no_ganFeat_loss=True, wt_l1=0.5, wt_vgg=2.0, wt_multiscale=0.25, wt_temporal=4.0),
}


def sams_hparams(**kw):
    base = dict(n_frames_total=3, n_frames_now=None, person_inputs=["agnostic", "densepose", "flow"], cloth_inputs=["cloth"],
                encoder_input="flow", flow_warp=True, activation="relu", fine_height=64, fine_width=48, is_train=True,
                norm_G="spectralspadesyncbatch3x3", ngf_base=2, ngf_pow_outer=3, ngf_pow_inner=5, ngf_pow_step=1, num_middle=2,
                attention_middle_indices=[], attention_decoder_indices=[], init_type="xavier", init_variance=0.02,
                netD_subarch="n_layer", num_D=2, n_layers_D=4, ndf=8, norm_D="spectralinstance", gan_mode="hinge", lr=1e-4,
                lr_D=3e-4, no_ganFeat_loss=False, wt_l1=1.0, wt_vgg=1.0, wt_multiscale=1.0, wt_temporal=1.0,
                display_count=10 ** 9, keep_epochs=5, decay_epochs=5)
    base.update(kw)
    return argparse.Namespace(**base)


def load_golden(tag):
    return np.load(os.path.join(HERE, "golden", f"sams_{tag}.npz"), allow_pickle=False)


def golden_shapes(g):
    return {str(k): tuple(int(x) for x in s.strip("()").split(",") if x.strip()) for k, s in zip(g["state_keys"], g["state_shapes"])}


STEP_NETS = ("generator", "multiscale_discriminator", "temporal_discriminator")


@contextlib.contextmanager
def kink_shift(delta_rel):
    """Move the kink of every ReLU / LeakyReLU the oracle evaluates from 0 to delta_rel * max|x| (either sign).

    Two fp32 implementations agree on a pre-activation to ~1e-6 of its tensor's magnitude; an element that close to 0
    can land on either side of the kink, and its whole gradient contribution changes by the slope difference.  Running
    the fp64 oracle with the kink shifted by +-delta brackets every such element: the spread of the resulting
    gradients is what any correct fp32 implementation may differ by, and it is exactly 0 when no element is that close."""
    import torch.nn.functional as F

    relu, leaky = F.relu, F.leaky_relu

    def shifted_relu(x, inplace=False):
        return torch.where(x > delta_rel * x.detach().abs().max(), x, torch.zeros_like(x))

    def shifted_leaky(x, negative_slope=0.01, inplace=False):
        return torch.where(x > delta_rel * x.detach().abs().max(), x, negative_slope * x)

    F.relu, F.leaky_relu = shifted_relu, shifted_leaky
    try:
        yield
    finally:
        F.relu, F.leaky_relu = relu, leaky


# The bracket's half-width, relative to the tensor's largest pre-activation.  Round 4 used 1e-5 - ten times the ~1e-6 two fp32
# evaluations differ by; round 5 tightened it to 2e-6 (the GPU's activations were measured within 4e-6 of the fp64 oracle's at
# full size, mostly far below) and the route tallies say how many tensors still need it.
KINK_DELTA = float(os.environ.get("SHINEON_KINK_DELTA", "2e-6"))


def kink_spread(sd, hp, batch, exact_steps, delta_rel=None):
    """Per step: {key: max |gradient(kink at +-delta) - gradient(kink at 0)|} from two more fp64 oracle runs."""
    delta_rel = KINK_DELTA if delta_rel is None else delta_rel
    spread = [dict() for _ in exact_steps]
    for sign in (+1.0, -1.0):
        with kink_shift(sign * delta_rel):
            steps, _, _ = oracle_three_steps(sd, hp, batch, torch.float64)
        for idx, (_, grads) in enumerate(steps):
            for k, g in grads.items():
                d = (g - exact_steps[idx][1][k]).abs().max().item()
                spread[idx][k] = max(spread[idx].get(k, 0.0), d)
    return spread


def oracle_three_steps(sd, hp, batch, dtype=torch.float32):
    """Generator / multiscale-D / temporal-D steps in Lightning's multi-optimizer order (only the current optimizer's
    parameters require grad).  Returns per step: logs, gradients {key: tensor}; plus the generated frames of step 0."""
    from oracle import sams_oracle as so

    sd = {k: (v.to(dtype).clone() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    batch = {k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in batch.items()}
    groups = so.optimizer_groups(sd)
    model = so.SamsOracle(sd, hp)
    out = []
    frames = None
    for idx, net in enumerate(STEP_NETS):
        for k, v in sd.items():
            if v.is_floating_point():
                v.requires_grad_(False)
                v.grad = None
        for k in groups[net]:
            sd[k].requires_grad_(True)
        loss, logs = (model.generator_step, model.multiscale_discriminator_step, model.temporal_discriminator_step)[idx](batch)
        loss.sum().backward()
        if idx == 0:
            frames = model.all_gen_frames.detach()
        out.append(({k: float(v.detach().sum()) for k, v in logs.items()},
                    {k: sd[k].grad.detach().clone() for k in groups[net] if sd[k].grad is not None}))
    return out, frames, sd
