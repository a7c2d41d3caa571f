"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

CPU restatement (plain PyTorch, NCHW, functional, state_dict-driven, dtype-generic so that it runs in fp32 and in
fp64) of the reference's SAMS-GAN path: `SamsGenerator`, SPADE / MultiSpade / AttentiveMultiSpade,
the spectral-norm power iteration, the multiscale and temporal PatchGAN discriminators, `GANLoss`, and the three
`SamsModel` training steps.

Parity pin: tests/test_oracle_golden.py::test_sams_* checks the three step losses, every logged scalar, generated
frames, updated buffers (u, v, running statistics) and gradients against tests/golden/sams_*.npz, which
tests/golden/make_golden.py::gen_sams produced by running the REFERENCE's own `SamsModel` (imported from
/root/reference through the shim) on the same procedural weights and synthetic batch.

Reference behaviour that looks odd but is restated as-is (each cited where it happens):
  * the generator's two adversarial terms are computed from the discriminator's prediction on the REAL frame
    (sams_model.py:299-303,334-338), so they carry no gradient to the generator;
  * the temporal discriminator returns its intermediate features and `GANLoss` averages the loss over ALL of them, not
    only the final patch map (discriminator.py:137-145, loss.py:92-101, sams_model.py:434-441);
  * previous frames are fed in rotated order `(f+1 .. f+n-1) mod n` and the previous label maps are
    `enc[:, n-1-f : -1]` (sams_model.py:255-270);
  * "syncbatch" is a plain per-process batch norm outside `DataParallelWithCallback`
    (sync_batchnorm/batchnorm.py:63-68) and, being `F.batch_norm`, never bumps `num_batches_tracked`;
  * the generator output has no tanh / sigmoid (sams_model.py:229-235);
  * one power iteration runs in EVERY training-mode forward, also inside `torch.no_grad()` generation
    (torch.nn.utils.spectral_norm, applied at spade.py:149-153 and normalization.py:24-25).
"""
import math
import re

import torch
import torch.nn.functional as F

from . import shineon_oracle as base

CHANNELS = {"agnostic": 4, "cocopose": 18, "densepose": 3, "cloth": 3, "flow": 2, "image": 3}

DEBUG_TAPS = None  # tools/ set this to a list to receive (name, tensor) for every intermediate of the SPADE blocks


def _tap(name, t):
    if DEBUG_TAPS is not None:
        if t.requires_grad:
            t.retain_grad()
        DEBUG_TAPS.append((name, t))
    return t


# ------------------------------------------------------------------------------------------------
# building blocks
# ------------------------------------------------------------------------------------------------
def resblock_activation(x, kind):
    """AnySpadeResBlock._get_activation_fn (sams/spade.py:182-192): "relu" means LeakyReLU(0.2) here."""
    if kind == "relu":
        return F.leaky_relu(x, 0.2)
    if kind in ("gelu", "swish", "sine"):
        return base.activation(x, kind)
    raise RuntimeError(f"The selected activation should be relu/gelu/swish/sine, not {kind}")


def spade_activation(x, kind):
    """SPADE._get_activation_fn (sams/spade.py:93-103): "relu" is a true ReLU inside the SPADE MLP."""
    if kind in ("relu", "gelu", "swish", "sine"):
        return base.activation(x, kind)
    raise RuntimeError(f"The selected activation should be relu/gelu/swish/sine, not {kind}")


def parse_spade_config(config_text):
    """SPADE.parse_config_text (sams/spade.py:36-59): "spade<norm><k>x<k>" -> (norm name, k)."""
    assert config_text.startswith("spade")
    m = re.search(r"spade(\D+)(\d)x\d", config_text)
    norm = m.group(1)
    if norm not in ("instance", "syncbatch", "batch"):
        raise ValueError("%s is not a recognized param-free norm type in SPADE" % norm)
    return norm, int(m.group(2))


def spectral_weight(sd, prefix, training, eps=1e-12):
    """torch.nn.utils.spectral_norm's pre-forward hook (one power iteration, dim 0): updates `weight_u` / `weight_v` in
    `sd` in place when training, and returns weight_orig / sigma with sigma = u^T W v differentiated through W only."""
    w = sd[prefix + ".weight_orig"]
    u, v = sd[prefix + ".weight_u"], sd[prefix + ".weight_v"]
    wm = w.reshape(w.shape[0], -1)
    if training:
        with torch.no_grad():
            v.copy_(F.normalize(torch.mv(wm.t(), u), dim=0, eps=eps))
            u.copy_(F.normalize(torch.mv(wm, v), dim=0, eps=eps))
        u, v = u.clone(), v.clone()
    sigma = torch.dot(u, torch.mv(wm, v))
    return w / sigma


def maybe_spectral_conv(sd, prefix, x, training, stride=1, padding=1):
    """A conv whose weight may or may not be spectrally normalised, decided by the keys present."""
    if prefix + ".weight_orig" in sd:
        w = spectral_weight(sd, prefix, training)
    else:
        w = sd[prefix + ".weight"]
    return F.conv2d(x, w, sd.get(prefix + ".bias"), stride=stride, padding=padding)


def _moments(x, dims):
    """Mean and biased variance over `dims`, accumulated in float64 and rounded to x.dtype: ATen's CPU batch / instance
    norm kernels accumulate in `acc_type<float, false>` = double, and the goldens come from that code."""
    xd = x.double()
    mean = xd.mean(dim=dims, keepdim=True)
    var = ((xd - mean) ** 2).mean(dim=dims, keepdim=True)
    return mean.to(x.dtype), var.to(x.dtype)


def instance_norm(x, eps=1e-5):
    """nn.InstanceNorm2d(affine=False) (normalization.py:43, spade.py:47-48)."""
    mean, var = _moments(x, (2, 3))
    return (x - mean) / torch.sqrt(var + eps)


def param_free_norm(sd, prefix, x, norm, training):
    """SPADE's parameter-free normalisation (sams/spade.py:65,80)."""
    if norm == "instance":
        return instance_norm(x)
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    if not training:
        return (x - rm[None, :, None, None]) / torch.sqrt(rv[None, :, None, None] + 1e-5)
    n = x.numel() / x.shape[1]
    mean, var = _moments(x, (0, 2, 3))
    mean, var = mean.reshape(-1), var.reshape(-1)
    with torch.no_grad():
        rm.mul_(0.9).add_(0.1 * mean)
        rv.mul_(0.9).add_(0.1 * var * n / (n - 1))
        if norm == "batch":  # nn.BatchNorm2d counts; the "syncbatch" fallback (plain F.batch_norm) does not
            sd[prefix + ".num_batches_tracked"] += 1
    return (x - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + 1e-5)


def spade(sd, prefix, x, segmap, hp, training):
    """SPADE.forward (sams/spade.py:77-91)."""
    norm, ks = parse_spade_config(hp.norm_G.replace("spectral", ""))
    normalized = _tap(prefix + ":normalized", param_free_norm(sd, prefix + ".param_free_norm", x, norm, training))
    seg = F.interpolate(segmap, size=x.shape[2:], mode="nearest")
    pw = ks // 2
    actv = spade_activation(F.conv2d(seg, sd[prefix + ".mlp_shared.0.weight"], sd[prefix + ".mlp_shared.0.bias"], padding=pw),
                            hp.activation)
    _tap(prefix + ":actv", actv)
    gamma = _tap(prefix + ":gamma", F.conv2d(actv, sd[prefix + ".mlp_gamma.weight"], sd[prefix + ".mlp_gamma.bias"], padding=pw))
    beta = _tap(prefix + ":beta", F.conv2d(actv, sd[prefix + ".mlp_beta.weight"], sd[prefix + ".mlp_beta.bias"], padding=pw))
    return _tap(prefix + ":out", normalized * (1 + gamma) + beta)


def multispade(sd, prefix, x, labelmaps, hp, training):
    """MultiSpade.forward (sams/multispade.py:49-65): the SPADEs run one after another in sorted key order."""
    for key, seg in sorted(labelmaps.items()):
        x = spade(sd, f"{prefix}.spade_layers.{key}", x, seg, hp, training)
    return x


def attentive_multispade(sd, prefix, x, labelmaps, hp, training):
    """AttentiveMultiSpade.forward (sams/attentive_multispade.py:34-50): parallel SPADEs on the same x, channel concat,
    SAGAN attention, 3x3 conv back to norm_nc + LeakyReLU() (default slope 0.01)."""
    _, ks = parse_spade_config(hp.norm_G.replace("spectral", ""))
    outs = [spade(sd, f"{prefix}.spade_layers.{key}", x, seg, hp, training) for key, seg in sorted(labelmaps.items())]
    attended = base.self_attention(torch.cat(outs, 1), sd, prefix + ".attention_layer")
    y = F.conv2d(attended, sd[prefix + ".mlp_final.0.weight"], sd[prefix + ".mlp_final.0.bias"], padding=ks // 2)
    return F.leaky_relu(y, 0.01)


def any_spade(sd, prefix, x, seg, hp, training):
    """Dispatch on what the state_dict holds: SPADE (encoder), MultiSpade or AttentiveMultiSpade."""
    if prefix + ".mlp_final.0.weight" in sd:
        return attentive_multispade(sd, prefix, x, seg, hp, training)
    if any(k.startswith(prefix + ".spade_layers.") for k in sd):
        if torch.is_tensor(seg):  # MultiSpade.try_fix_labelmap_dict (multispade.py:67-77)
            keys = sorted({k[len(prefix) + 14:].split(".")[0] for k in sd if k.startswith(prefix + ".spade_layers.")})
            if len(keys) != 1:
                raise ValueError("a single Tensor was passed but there are several spade layers")
            seg = {keys[0]: seg}
        return multispade(sd, prefix, x, seg, hp, training)
    return spade(sd, prefix, x, seg, hp, training)


def spade_resblock(sd, prefix, x, seg, hp, training):
    """AnySpadeResBlock.forward (sams/spade.py:165-180)."""
    learned = (prefix + ".conv_s.weight_orig" in sd) or (prefix + ".conv_s.weight" in sd)
    if learned:
        x_s = maybe_spectral_conv(sd, prefix + ".conv_s", any_spade(sd, prefix + ".norm_s", x, seg, hp, training), training, padding=0)
    else:
        x_s = x
    dx = maybe_spectral_conv(sd, prefix + ".conv_0", resblock_activation(any_spade(sd, prefix + ".spade_0", x, seg, hp, training),
                                                                         hp.activation), training)
    dx = maybe_spectral_conv(sd, prefix + ".conv_1", resblock_activation(any_spade(sd, prefix + ".spade_1", dx, seg, hp, training),
                                                                         hp.activation), training)
    return x_s + dx


def generator_layout(hp):
    """The layer lists SamsGenerator.__init__ builds (sams_generator.py:123-212), as tuples
    ("conv" | "block" | "down" | "up", index in its ModuleList)."""
    enc, dec = [("conv", 0)], []
    i = 1
    out_feat = int(hp.ngf_base ** hp.ngf_pow_outer)
    outer, inner = out_feat, int(hp.ngf_base ** hp.ngf_pow_inner)
    for p in range(hp.ngf_pow_outer, hp.ngf_pow_inner, hp.ngf_pow_step):
        out_feat = int(hp.ngf_base ** (p + hp.ngf_pow_step))
        enc += [("block", i), ("down", i + 1)]
        i += 2
    if out_feat != inner:
        enc += [("block", i), ("down", i + 1)]
    i = 0
    for p in range(hp.ngf_pow_inner, hp.ngf_pow_outer, -hp.ngf_pow_step):
        out_feat = int(hp.ngf_base ** (p - hp.ngf_pow_step))
        dec += [("up", i), ("block", i + 1)]
        i += 2
    if out_feat != outer:
        dec += [("up", i), ("block", i + 1)]
        i += 2
    dec.append(("conv", i))
    return enc, list(range(hp.num_middle)), dec


def generator_forward(sd, prev_frames, prev_labelmaps, current_labelmaps, hp, training, prefix="generator", taps=None):
    """SamsGenerator.forward (sams_generator.py:240-291).  prev_frames / prev_labelmaps: (b, n-1, c, h, w).
    taps: optional list that receives (name, activation) after every layer (tests localise a mismatch with it)."""
    def tap(name, t):
        if taps is not None:
            if t.requires_grad:
                t.retain_grad()
            taps.append((name, t))
        return t

    b, n, c, h, w = prev_frames.shape
    x = prev_frames.reshape(b, n * c, h, w)
    prev_maps = prev_labelmaps.reshape(b, -1, h, w)
    enc, mid, dec = generator_layout(hp)
    for kind, i in enc:
        if kind == "conv":
            x = F.conv2d(x, sd[f"{prefix}.encode_layers.{i}.weight"], sd[f"{prefix}.encode_layers.{i}.bias"], padding=1)
        elif kind == "block":
            x = spade_resblock(sd, f"{prefix}.encode_layers.{i}", x, prev_maps, hp, training)
        else:  # nn.Upsample(scale_factor=0.5), default mode "nearest" (sams_generator.py:299)
            x = F.interpolate(x, scale_factor=0.5, mode="nearest")
        tap(f"encode_layers.{i}", x)
    for i in mid:
        x = tap(f"middle_layers.{i}", spade_resblock(sd, f"{prefix}.middle_layers.{i}", x, current_labelmaps, hp, training))
    for kind, i in dec:
        if kind == "conv":
            x = F.conv2d(x, sd[f"{prefix}.decode_layers.{i}.weight"], sd[f"{prefix}.decode_layers.{i}.bias"], padding=1)
        elif kind == "block":
            x = spade_resblock(sd, f"{prefix}.decode_layers.{i}", x, current_labelmaps, hp, training)
        else:
            x = F.interpolate(x, scale_factor=2, mode="nearest")
        tap(f"decode_layers.{i}", x)
    return x


# ------------------------------------------------------------------------------------------------
# discriminators and the GAN loss
# ------------------------------------------------------------------------------------------------
def nlayer_discriminator(sd, prefix, x, hp, training):
    """NLayerDiscriminator.forward (discriminator.py:94-145): 4x4 convs with padding 2; first and last layers plain,
    the middle ones norm_D (default spectral norm + InstanceNorm, bias dropped); LeakyReLU(0.2) after all but the last."""
    if not hp.norm_D.startswith("spectral"):
        # normalization.py:24-28 only binds subnorm_type inside the `startswith("spectral")` branch
        raise UnboundLocalError("local variable 'subnorm_type' referenced before assignment")
    subnorm = hp.norm_D[len("spectral"):]
    feats = []
    n_layers = hp.n_layers_D
    for n in range(n_layers + 1):
        first, last = n == 0, n == n_layers
        if first or last:
            x = F.conv2d(x, sd[f"{prefix}.model{n}.0.weight"], sd[f"{prefix}.model{n}.0.bias"], stride=2 if first else 1, padding=2)
        else:
            stride = 1 if n == n_layers - 1 else 2
            if subnorm in ("", "none"):
                x = maybe_spectral_conv(sd, f"{prefix}.model{n}.0", x, training, stride=stride, padding=2)
            else:
                x = maybe_spectral_conv(sd, f"{prefix}.model{n}.0.0", x, training, stride=stride, padding=2)
                if subnorm == "instance":
                    x = instance_norm(x)
                elif subnorm in ("batch", "sync_batch"):
                    p = f"{prefix}.model{n}.0.1"
                    xn = param_free_norm(sd, p, x, "batch" if subnorm == "batch" else "syncbatch", training)
                    x = xn * sd[p + ".weight"][None, :, None, None] + sd[p + ".bias"][None, :, None, None]
                else:
                    raise ValueError("normalization layer %s is not recognized" % subnorm)
        if not last:
            x = F.leaky_relu(x, 0.2)
        feats.append(x)
    return feats if not hp.no_ganFeat_loss else feats[-1]


def multiscale_discriminator(sd, prefix, x, hp, training):
    """MultiscaleDiscriminator.forward (discriminator.py:57-75)."""
    result = []
    for i in range(hp.num_D):
        out = nlayer_discriminator(sd, f"{prefix}.discriminator_{i}", x, hp, training)
        result.append(out if not hp.no_ganFeat_loss else [out])
        x = F.avg_pool2d(x, kernel_size=3, stride=2, padding=[1, 1], count_include_pad=False)
    return result


def gan_loss_single(x, mode, target_is_real, for_discriminator):
    """GANLoss.loss (loss.py:58-88) with the default labels 1.0 / 0.0."""
    if mode == "original":
        return F.binary_cross_entropy_with_logits(x, torch.full_like(x, 1.0 if target_is_real else 0.0))
    if mode == "ls":
        return F.mse_loss(x, torch.full_like(x, 1.0 if target_is_real else 0.0))
    if mode == "hinge":
        if for_discriminator:
            z = torch.zeros_like(x)
            return -torch.mean(torch.min(x - 1, z)) if target_is_real else -torch.mean(torch.min(-x - 1, z))
        assert target_is_real, "The generator's hinge loss must be aiming for real"
        return -torch.mean(x)
    if mode == "w":
        return -x.mean() if target_is_real else x.mean()
    raise AssertionError(f"Unexpected gan_mode = {mode}")


def gan_loss(pred, mode, target_is_real, for_discriminator=True):
    """GANLoss.__call__ (loss.py:90-103): a list is averaged over its entries, a list of lists uses each inner list's
    last entry; the result of the list branch has shape (1,)."""
    if isinstance(pred, list):
        loss = 0
        for p in pred:
            if isinstance(p, list):
                p = p[-1]
            loss = loss + gan_loss_single(p, mode, target_is_real, for_discriminator).reshape(1)
        return loss / len(pred)
    return gan_loss_single(pred, mode, target_is_real, for_discriminator)


def split_predictions(pred):
    """sams_model.py:428-449."""
    if isinstance(pred, list):
        fake, real = [], []
        for p in pred:
            if torch.is_tensor(p):
                fake.append(p[: p.size(0) // 2])
                real.append(p[p.size(0) // 2:])
            else:
                fake.append([t[: t.size(0) // 2] for t in p])
                real.append([t[t.size(0) // 2:] for t in p])
        return fake, real
    return pred[: pred.size(0) // 2], pred[pred.size(0) // 2:]


# ------------------------------------------------------------------------------------------------
# SamsModel
# ------------------------------------------------------------------------------------------------
class SamsOracle:
    """The three training steps of SamsModel (sams_model.py:147-383) over a reference-layout state_dict.  Buffers in `sd`
    (running statistics, u / v of the power iteration) are updated in place exactly when the reference updates them."""

    def __init__(self, sd, hp):
        self.sd, self.hp = sd, hp
        self.n_total = hp.n_frames_total
        self.n_now = hp.n_frames_now if getattr(hp, "n_frames_now", None) else self.n_total
        self.inputs = list(hp.person_inputs) + list(hp.cloth_inputs)
        self.all_gen_frames = None

    # -- generation ------------------------------------------------------------------------------
    def prev_frames_and_maps(self, batch, f, frames):
        """get_prev_frames_and_maps (sams_model.py:241-272)."""
        enc = batch[self.hp.encoder_input]
        n = self.n_total
        if n == 1:
            raise IndexError("SamsModel needs n_frames_total > 1 (the reference indexes a frames axis, sams_model.py:220)")
        n_prev = n - 1
        idx = [(i + 1) % n for i in range(f, f + n_prev)]
        prev_frames = torch.stack([frames[i] for i in idx], 1).detach()
        b, _, c, h, w = enc.shape
        start = n_prev - f
        prev_maps = torch.cat((torch.zeros(b, start, c, h, w, dtype=enc.dtype), enc[:, start:-1]), 1)
        return prev_frames, prev_maps

    def generate_n_frames(self, batch, training=True):
        """generate_n_frames (sams_model.py:204-239)."""
        hp = self.hp
        image = batch["image"]
        frames = [torch.zeros_like(image[:, 0]) for _ in range(self.n_total)]
        fake = maps_now = None
        for f in range(self.n_total - self.n_now, self.n_total):
            maps_now = {k: batch[k][:, f] for k in self.inputs}
            prev_frames, prev_maps = self.prev_frames_and_maps(batch, f, frames)
            out = generator_forward(self.sd, prev_frames, prev_maps, maps_now, hp, training)
            fake, weight_mask = out[:, :3], out[:, 3:]
            if hp.flow_warp:
                last = frames[f - 1] if f > 0 else torch.zeros_like(frames[f])
                warped = base.resample2d(last, batch["flow"][:, f].contiguous())
                fake = (1 - weight_mask) * warped + weight_mask * fake
            frames[f] = fake
        return fake, maps_now, torch.stack(frames, 1)

    def mask_unused_frames(self, t):
        """sams_model.py:346-361."""
        n_mask = self.n_total - self.n_now
        return torch.cat((torch.zeros_like(t[:, :n_mask]), t[:, n_mask:]), 1)

    def discriminate(self, which, semantics, fake, real, training=True):
        """sams_model.py:385-403: fake and real halves go through the discriminator as ONE batch."""
        both = torch.cat([torch.cat([semantics, fake], 1), torch.cat([semantics, real], 1)], 0)
        if which == "multiscale":
            out = multiscale_discriminator(self.sd, "multiscale_discriminator", both, self.hp, training)
        else:
            out = nlayer_discriminator(self.sd, "temporal_discriminator", both, self.hp, training)
        return split_predictions(out)

    # -- losses ------------------------------------------------------------------------------------
    def multiscale_adversarial_loss(self, batch, for_discriminator, training=True):
        """sams_model.py:274-309."""
        if not for_discriminator:
            fake, maps_now, frames = self.generate_n_frames(batch, training)
            self.all_gen_frames = frames
        else:
            with torch.no_grad():
                fake, maps_now, frames = self.generate_n_frames(batch, training)
            fake = fake.detach().requires_grad_()
            self.all_gen_frames = frames.detach().requires_grad_()
        semantics = torch.cat(tuple(maps_now.values()), 1)
        pred_fake, pred_real = self.discriminate("multiscale", semantics, fake, batch["image"][:, -1], training)
        loss_real = gan_loss(pred_real, self.hp.gan_mode, True, for_discriminator)
        if not for_discriminator:
            return loss_real
        loss_fake = gan_loss(pred_fake, self.hp.gan_mode, False, for_discriminator)
        return (loss_fake + loss_real) / 2, loss_real, loss_fake

    def temporal_adversarial_loss(self, batch, for_discriminator, training=True):
        """sams_model.py:311-344."""
        reals = self.mask_unused_frames(batch["image"])
        b, _, _, h, w = reals.shape
        reals = reals.reshape(b, -1, h, w)
        fakes = self.all_gen_frames.reshape(b, -1, h, w)
        semantics = self.mask_unused_frames(batch[self.hp.encoder_input]).reshape(b, -1, h, w)
        pred_fake, pred_real = self.discriminate("temporal", semantics, fakes, reals, training)
        loss_real = gan_loss(pred_real, self.hp.gan_mode, True, for_discriminator)
        if not for_discriminator:
            return loss_real
        loss_fake = gan_loss(pred_fake, self.hp.gan_mode, False, for_discriminator)
        return (loss_fake + loss_real) / 2, loss_real, loss_fake

    # -- steps -------------------------------------------------------------------------------------
    def generator_step(self, batch, val=False):
        """generator_step (sams_model.py:173-202).  Returns (loss to minimise or checkpoint_on, logs)."""
        hp = self.hp
        training = not val
        adv_multi = self.multiscale_adversarial_loss(batch, False, training) * hp.wt_multiscale
        adv_temporal = self.temporal_adversarial_loss(batch, False, training) * hp.wt_temporal
        truth = batch["image"][:, -1]
        fake = self.all_gen_frames[:, -1]
        l1 = F.l1_loss(fake, truth) * hp.wt_l1
        vgg = base.vgg_loss(self.sd, fake, truth, prefix="criterion_VGG.vgg") * hp.wt_vgg
        loss = l1 + vgg + adv_multi + adv_temporal
        v = "val_" if val else ""
        logs = {f"{v}loss": loss, f"{v}loss/G/adv_multiscale": adv_multi, f"{v}loss/G/adv_temporal": adv_temporal,
                f"{v}loss/G/l1+vgg": l1 + vgg, f"{v}loss/G/l1": l1, f"{v}loss/G/vgg": vgg}
        return (l1 + vgg if val else loss), logs

    def multiscale_discriminator_step(self, batch):
        """sams_model.py:363-372."""
        loss, real, fake = self.multiscale_adversarial_loss(batch, True)
        return loss, {"loss/D/multi": loss, "loss/D/multi_fake": fake, "loss/D/multi_real": real}

    def temporal_discriminator_step(self, batch):
        """sams_model.py:374-383 (uses the frames the multiscale discriminator step generated)."""
        loss, real, fake = self.temporal_adversarial_loss(batch, True)
        return loss, {"loss/D/temporal": loss, "loss/D/temporal_fake": fake, "loss/D/temporal_real": real}


def optimizer_groups(sd):
    """configure_optimizers (sams_model.py:130-145): key prefixes of the three Adam parameter sets."""
    floats = [k for k, v in sd.items() if v.is_floating_point()]
    is_param = [k for k in floats if not re.search(r"(running_mean|running_var|weight_u|weight_v)$", k)]
    return {
        "generator": [k for k in is_param if k.startswith("generator.")],
        "multiscale_discriminator": [k for k in is_param if k.startswith("multiscale_discriminator.")],
        "temporal_discriminator": [k for k in is_param if k.startswith("temporal_discriminator.")],
    }


def xavier_std(shape, gain):
    """init.xavier_normal_ (used by BaseNetwork.init_weights, base_network.py:58): std = gain * sqrt(2 / (fan_in + fan_out))."""
    rf = 1
    for s in shape[2:]:
        rf *= s
    return gain * math.sqrt(2.0 / (shape[1] * rf + shape[0] * rf))
