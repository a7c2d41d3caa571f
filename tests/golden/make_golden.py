"""Generates tests/golden/*.npz by running the REFERENCE's own modules (imported from /root/reference
through an in-memory shim for its missing third-party imports) on procedural weights and synthetic inputs.

Runs only in the build container (needs /root/reference); the committed .npz files are what travels.
Nothing of the reference's source is copied: the shim only provides stand-ins for torchvision /
pytorch_lightning / tensorboard / the empty flownet2 submodule so that `import models.warp_model` works.

    python tests/golden/make_golden.py
"""
import argparse
import collections
import collections.abc
import os
import sys
import types

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import shineon_oracle as oracle  # noqa: E402  (only for the Resample2d stand-in)
from oracle.procedural import procedural_state_dict, shapes_of  # noqa: E402

import shineon_virtual_tryon_amd  # noqa: E402,F401
from shineon_virtual_tryon_amd.data import synthetic_batch  # noqa: E402  (pure numpy generator)


# ------------------------------------------------------------------------------------------------
# shim
# ------------------------------------------------------------------------------------------------
def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_shim():
    collections.Iterable = collections.abc.Iterable

    class _Dummy:
        def __init__(self, *a, **k):
            pass

        def __call__(self, x):
            return x

    tv = _mod("torchvision")
    tv.transforms = _mod("torchvision.transforms", CenterCrop=_Dummy, Normalize=_Dummy, Compose=_Dummy,
                         ToTensor=_Dummy, Resize=_Dummy)
    tv.utils = _mod("torchvision.utils", save_image=lambda *a, **k: None, make_grid=lambda *a, **k: None)

    def vgg19(pretrained=False):
        cfg = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M",
               512, 512, 512, 512, "M"]
        layers, cin = [], 3
        for v in cfg:
            if v == "M":
                layers.append(nn.MaxPool2d(2, 2))
            else:
                layers += [nn.Conv2d(cin, v, 3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        return types.SimpleNamespace(features=nn.Sequential(*layers))

    tv.models = _mod("torchvision.models", vgg19=vgg19)

    class Resample2d(nn.Module):
        def forward(self, a, b):
            return oracle.resample2d(a, b)

    _mod("models.flownet2_pytorch")
    _mod("models.flownet2_pytorch.utils")
    _mod("models.flownet2_pytorch.utils.flow_utils", flow2img=lambda *a: None, readFlow=lambda *a: None)
    _mod("models.flownet2_pytorch.networks")
    _mod("models.flownet2_pytorch.networks.resample2d_package")
    _mod("models.flownet2_pytorch.networks.resample2d_package.resample2d", Resample2d=Resample2d)

    class _Result:
        def __init__(self, minimize=None, checkpoint_on=None, **k):
            self.minimize, self.checkpoint_on, self.logs = minimize, checkpoint_on, {}

        def log(self, name, value, **k):
            self.logs[name] = value

    class LightningModule(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
            self.global_step = 1  # never 0: skips visualize()

    pl = _mod("pytorch_lightning", LightningModule=LightningModule, TrainResult=_Result, EvalResult=_Result,
              Trainer=object, Callback=object)
    pl.callbacks = _mod("pytorch_lightning.callbacks", ModelCheckpoint=object, Callback=object)
    _mod("torch.utils.tensorboard", SummaryWriter=object)
    nn.Module.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)


def hp_namespace(**kw):
    base = dict(n_frames_total=1, person_inputs=["agnostic", "densepose"], cloth_inputs=["cloth"], is_train=True, ngf=64,
                grid_size=5, fine_height=256, fine_width=192, self_attn=False, num_attn=2, flow_warp=False,
                activation=None, display_count=10 ** 9, pen_flow_mask=1.0)
    base.update(kw)
    return argparse.Namespace(**base)


def strided(t, s=8):
    return t.detach()[..., ::s, ::s].contiguous().numpy()


def checksums(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()], dtype=np.float64)


def grad_checksums(module):
    return {k: checksums(p.grad) for k, p in module.named_parameters() if p.grad is not None}


def grad_samples(module, stride=97):
    """Every `stride`-th element (flattened OIHW order) of every parameter gradient: ties the oracle's (and the HIP path's)
    gradients to the reference's ELEMENT-WISE, not only through three checksums per tensor."""
    return {k: p.grad.detach().contiguous().reshape(-1)[::stride].clone().numpy()
            for k, p in module.named_parameters() if p.grad is not None}


# ------------------------------------------------------------------------------------------------
# generators
# ------------------------------------------------------------------------------------------------
def gen_warp(out):
    from models.warp_model import WarpModel

    hp = hp_namespace(person_inputs=["agnostic", "cocopose"])
    torch.manual_seed(0)
    model = WarpModel(hp)
    model.load_state_dict(procedural_state_dict(shapes_of(model.state_dict())))
    model.train()
    batch = synthetic_batch(2, "cpu", smooth=True)
    res = model.training_step(batch, 0)
    res.minimize.backward()
    person = torch.cat([batch[k] for k in hp.person_inputs], 1)
    model2 = WarpModel(hp)
    model2.load_state_dict(procedural_state_dict(shapes_of(model2.state_dict())))
    model2.train()
    with torch.no_grad():
        grid, theta = model2(person, batch["cloth"])
    g = model.gridGen
    data = {
        "loss": np.float64(res.minimize.item()), "theta": theta.numpy(), "grid_s8": strided(grid.permute(0, 3, 1, 2)),
        "grid_cs": checksums(grid), "warped_cloth_s8": strided(model.warped_cloth), "warped_cloth_cs": checksums(model.warped_cloth),
        "grad_linear_weight": model.regression.linear.weight.grad.numpy(),
        "grad_linear_bias": model.regression.linear.bias.grad.numpy(),
        "bn_rm_A2": model.extractionA.model[2].running_mean.numpy(), "bn_rv_A2": model.extractionA.model[2].running_var.numpy(),
        "bn_rm_R10": model.regression.conv[10].running_mean.numpy(), "bn_rv_R10": model.regression.conv[10].running_var.numpy(),
        "Li": g.Li[0].numpy(), "P_X_base": g.P_X_base.numpy(), "P_Y_base": g.P_Y_base.numpy(),
        "grid_X_row": g.grid_X[0, 0, :, 0, 0].numpy() if g.grid_X.dim() == 5 else g.grid_X[0, 0, :, 0].numpy(),
        "grid_Y_col": g.grid_Y[0, :, 0, 0, 0].numpy() if g.grid_Y.dim() == 5 else g.grid_Y[0, :, 0, 0].numpy(),
        "state_keys": np.array(list(model.state_dict().keys())),
        "state_shapes": np.array([str(tuple(v.shape)) for v in model.state_dict().values()]),
    }
    for k, v in grad_checksums(model).items():
        data["gcs:" + k] = v
    for k, v in grad_samples(model).items():
        data["gs97:" + k] = v
    np.savez_compressed(os.path.join(out, "warp_model.npz"), **data)
    print("warp_model.npz loss", data["loss"])


def gen_unet(out):
    from models.unet_mask_model import UnetMaskModel

    for tag, kw in (("plain", dict()), ("gelu", dict(activation="gelu")), ("attn", dict(self_attn=True)),
                    ("attn_gelu", dict(self_attn=True, activation="gelu"))):
        hp = hp_namespace(**kw)
        torch.manual_seed(0)
        model = UnetMaskModel(hp)
        model.load_state_dict(procedural_state_dict(shapes_of(model.state_dict())))
        model.train()
        batch = synthetic_batch(2, "cpu", smooth=True)
        res = model.training_step(batch, 0)
        res.minimize.backward()
        data = {"state_keys": np.array(list(model.state_dict().keys())),
                "state_shapes": np.array([str(tuple(v.shape)) for v in model.state_dict().values()])}
        for k, v in res.logs.items():
            data["log:" + k] = np.float64(v.item())
        for name, t in (("p_rendered", model.p_rendereds[0]), ("tryon_mask", model.tryon_masks[0]), ("p_tryon", model.p_tryons[0])):
            data[name + "_s8"] = strided(t)
            data[name + "_cs"] = checksums(t)
        for k, v in grad_checksums(model.unet).items():
            data["gcs:unet." + k] = v
        for k, v in grad_samples(model.unet).items():
            data["gs97:unet." + k] = v
        np.savez_compressed(os.path.join(out, f"unet_mask_{tag}.npz"), **data)
        print(f"unet_mask_{tag}.npz", {k: float(v) for k, v in data.items() if k.startswith("log:")})


def gen_unet_nframes(out):
    """n_frames_total = 3 with flow warping: ngf = int(64 * (ln 3 + 1)) = 134 (channel counts that are NOT multiples
    of 4), 5 output channels per frame, previous generated frame warped by Resample2d (the oracle's stand-in, since
    the flownet2 submodule is empty) and blended with the flow mask.  128 x 64 images keep the fixture small."""
    from models.unet_mask_model import UnetMaskModel

    hp = hp_namespace(n_frames_total=3, flow_warp=True, activation="gelu", fine_height=128, fine_width=64)
    torch.manual_seed(0)
    model = UnetMaskModel(hp)
    model.load_state_dict(procedural_state_dict(shapes_of(model.state_dict())))
    model.train()
    batch = synthetic_batch(2, "cpu", height=128, width=64, n_frames=3, smooth=True)
    res = model.training_step(batch, 0)
    res.minimize.backward()
    data = {"state_keys": np.array(list(model.state_dict().keys())),
            "state_shapes": np.array([str(tuple(v.shape)) for v in model.state_dict().values()])}
    for k, v in res.logs.items():
        data["log:" + k] = np.float64(v.item())
    for name, ts in (("p_rendered", model.p_rendereds), ("tryon_mask", model.tryon_masks), ("p_tryon", model.p_tryons),
                     ("flow_mask", model.flow_masks)):
        t = torch.cat(list(ts), 1)
        data[name + "_s4"] = strided(t, 4)
        data[name + "_cs"] = checksums(t)
    for k, v in grad_checksums(model.unet).items():
        data["gcs:unet." + k] = v
    for k, v in grad_samples(model.unet, 397).items():
        data["gs397:unet." + k] = v
    np.savez_compressed(os.path.join(out, "unet_mask_n3_flow.npz"), **data)
    print("unet_mask_n3_flow.npz", {k: float(v) for k, v in data.items() if k.startswith("log:")})


def gen_ops(out):
    """Per-op goldens from the reference's nn.Modules at small sizes (full tensors)."""
    from models.networks.attention.sagan import SelfAttention
    from models.networks.cpvton.unet import UnetSkipConnectionBlock
    from models.networks.cpvton.warp import FeatureCorrelation, FeatureL2Norm, TpsGridGen

    rng = np.random.default_rng(7)
    data = {}
    # SelfAttention, gamma = 0.7, N = 12 / 48 / 192
    for (h, w) in ((4, 3), (8, 6), (16, 12)):
        sa = SelfAttention(64, "relu")
        sa.load_state_dict(procedural_state_dict(shapes_of(sa.state_dict()), seed=11))
        x = torch.from_numpy(rng.normal(size=(2, 64, h, w)).astype(np.float32))
        data[f"sa_x_{h}x{w}"] = x.numpy()
        data[f"sa_y_{h}x{w}"] = sa(x).detach().numpy()
    # L2 norm + correlation
    fa = torch.from_numpy(rng.normal(size=(2, 32, 16, 12)).astype(np.float32))
    fb = torch.from_numpy(rng.normal(size=(2, 32, 16, 12)).astype(np.float32))
    na, nb = FeatureL2Norm()(fa), FeatureL2Norm()(fb)
    data.update(l2_x=fa.numpy(), l2_y=na.numpy(), corr_a=na.numpy(), corr_b=nb.numpy(),
                corr_y=FeatureCorrelation()(na, nb).contiguous().numpy())
    # TPS: theta = 0 and random theta
    tps = TpsGridGen(256, 192, grid_size=5)
    th = torch.from_numpy((0.3 * rng.uniform(-1, 1, size=(2, 50))).astype(np.float32))
    data.update(tps_theta=th.numpy(), tps_grid_s4=tps(th)[:, ::4, ::4].contiguous().numpy(),
                tps_grid0_s4=tps(torch.zeros(1, 50))[:, ::4, ::4].contiguous().numpy())
    # skip-connection blocks (small C): innermost / middle x {None, gelu}, InstanceNorm
    for act in (None, "gelu"):
        inner = UnetSkipConnectionBlock(16, 16, innermost=True, norm_layer=nn.InstanceNorm2d, activation=act)
        mid = UnetSkipConnectionBlock(8, 16, submodule=inner, norm_layer=nn.InstanceNorm2d, activation=act)
        mid.load_state_dict(procedural_state_dict(shapes_of(mid.state_dict()), seed=13))
        x = torch.from_numpy(rng.normal(size=(2, 8, 16, 12)).astype(np.float32))
        data[f"blk_x_{act}"] = x.numpy()
        data[f"blk_y_{act}"] = mid(x.clone()).detach().numpy()
        data[f"blk_keys_{act}"] = np.array(list(mid.state_dict().keys()))
    np.savez_compressed(os.path.join(out, "ops.npz"), **data)
    print("ops.npz", len(data), "arrays")


def gen_init(out):
    """init_weights (models/networks/__init__.py:52-96) under a fixed torch seed: every draw comes from the global
    generator in nn.Module.apply order, so a re-implementation that walks the same module tree reproduces the values.
    Records checksums + the first 32 values of every tensor in the state_dict after `manual_seed(1234); init_weights`."""
    from models.networks import init_weights
    from models.networks.cpvton.unet import UnetGenerator
    from models.networks.cpvton.warp import FeatureExtraction, FeatureRegression

    data = {}

    def record(tag, net):
        sd = net.state_dict()
        data[f"{tag}:keys"] = np.array(list(sd.keys()))
        for k, v in sd.items():
            if v.is_floating_point():
                data[f"{tag}:cs:{k}"] = checksums(v)
                data[f"{tag}:head:{k}"] = v.detach().reshape(-1)[:32].numpy().copy()

    for init_type in ("normal", "xavier", "kaiming"):
        torch.manual_seed(99)
        unet = UnetGenerator(10, 4, 6, 2, ngf=8, norm_layer=nn.InstanceNorm2d, use_self_attn=True, activation="gelu")
        torch.manual_seed(1234)
        init_weights(unet, init_type)
        record(f"unet_{init_type}", unet)
    torch.manual_seed(99)
    fe = FeatureExtraction(22, ngf=64, n_layers=3, norm_layer=nn.BatchNorm2d)
    torch.manual_seed(1234)
    init_weights(fe.model, "normal")
    record("fe_normal", fe)
    # FeatureRegression is NOT passed through init_weights by the reference (warp.py:70-99): PyTorch defaults.  Applying
    # it by hand shows the BatchNorm2d / Conv rules on a net that has both.
    torch.manual_seed(99)
    fr = FeatureRegression(input_nc=192, output_dim=50)
    torch.manual_seed(1234)
    init_weights(fr, "normal") if hasattr(torch.nn.init, "normal") else None
    record("fr_normal", fr)
    np.savez_compressed(os.path.join(out, "init_weights.npz"), **data)
    print("init_weights.npz", len(data), "arrays")


def _real_transforms():
    """Working stand-ins for the three torchvision transforms the dataset code composes (torchvision is absent): ToTensor
    (PIL uint8 -> CHW float / 255), Normalize (in-place (t - mean) / std, which raises on a channel-count mismatch exactly
    like torchvision's in-place broadcast does - the reference relies on that exception to fall back to its gray
    transform), CenterCrop (identity at the native size)."""
    from PIL import Image

    class ToTensor:
        def __call__(self, pic):
            a = np.array(pic, copy=True)
            if a.ndim == 2:
                a = a[:, :, None]
            return torch.from_numpy(a).permute(2, 0, 1).contiguous().to(torch.float32).div(255)

    class Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = torch.tensor(mean, dtype=torch.float32), torch.tensor(std, dtype=torch.float32)

        def __call__(self, t):
            t = t.clone()
            return t.sub_(self.mean[:, None, None]).div_(self.std[:, None, None])

    class CenterCrop:
        def __init__(self, size):
            self.size = size

        def __call__(self, img):
            th, tw = self.size
            w, h = img.size if isinstance(img, Image.Image) else (img.shape[-1], img.shape[-2])
            if (h, w) == (th, tw):
                return img
            top, left = int(round((h - th) / 2.0)), int(round((w - tw) / 2.0))
            return img.crop((left, top, left + tw, top + th))

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    return ToTensor, Normalize, CenterCrop, Compose


def synthetic_raw_sample(seed, h, w):
    """Raw dataset inputs: a blocky LIP parse map, a low-entropy RGB image, 18 keypoints (some missing, some at the
    borders, fractional), a .flo payload."""
    rng = np.random.default_rng(seed)
    labels = rng.integers(0, 20, size=(h // 8 + 1, w // 8 + 1)).astype(np.uint8)
    labels[rng.random(labels.shape) < 0.35] = 0
    parse = np.kron(labels, np.ones((8, 8), np.uint8))[:h, :w]
    parse = np.roll(parse, (3, 5), (0, 1))
    tiles = rng.integers(0, 256, size=(h // 4 + 1, w // 4 + 1, 3)).astype(np.uint8)
    image = np.kron(tiles, np.ones((4, 4, 1), np.uint8))[:h, :w]
    kp = np.zeros((18, 3), np.float64)
    kp[:, 0] = rng.uniform(-4, w + 4, 18)
    kp[:, 1] = rng.uniform(-4, h + 4, 18)
    kp[:, 2] = rng.uniform(0, 1, 18)
    kp[3] = 0.0                      # undetected joint
    kp[5, :2] = (0.6, h / 2)         # x <= 1: skipped
    kp[7, :2] = (w - 1.25, 2.75)     # clipped at two borders
    kp[9, :2] = (37.0, 41.0)         # integer coordinates
    flow = rng.normal(0, 2, size=(h, w, 2)).astype(np.float32)
    flow = np.round(flow * 4) / 4    # low entropy (keeps the fixture small), still exercises (x - 0.5) / 0.5
    return parse, image, kp, flow.astype(np.float32)


def gen_dataprep(out):
    """Dataset-side tensor preparation through the reference's own TryonDataset methods."""
    import tempfile

    from PIL import Image

    from oracle import dataprep_oracle as dpo
    ToTensor, Normalize, CenterCrop, Compose = _real_transforms()
    sys.modules["models.flownet2_pytorch.utils.flow_utils"].readFlow = lambda path: dpo.read_flo(open(path, "rb").read())
    import datasets.tryon_dataset as td
    from datasets.util import segment_cloths_from_image

    td.readFlow = sys.modules["models.flownet2_pytorch.utils.flow_utils"].readFlow
    data = {}
    for tag, (h, w), seed in (("full", (256, 192), 5), ("small", (80, 48), 6)):
        parse, image, kp, flow = synthetic_raw_sample(seed, h, w)
        crop = CenterCrop((h, w))
        ds = types.SimpleNamespace(
            opt=types.SimpleNamespace(visualize_flow=False, fine_height=h, fine_width=w), fine_height=h, fine_width=w, radius=5,
            center_crop=crop, to_tensor_and_norm_rgb=Compose([crop, ToTensor(), Normalize((0.5,) * 3, (0.5,) * 3)]),
            to_tensor_and_norm_gray=Compose([crop, ToTensor(), Normalize([0.5], [0.5])]),
            flow_norm=Normalize((0.5, 0.5), (0.5, 0.5)))
        im = ds.to_tensor_and_norm_rgb(Image.fromarray(image))
        head = td.TryonDataset.get_person_head(ds, im, parse)
        cloth = segment_cloths_from_image(im, parse)
        sil = td.TryonDataset.get_person_body_silhouette(ds, parse)
        ds.convert = None
        pm, vis = td.TryonDataset.convert_pose_data_to_pose_map_and_vis(ds, kp)
        pm0, vis0 = td.TryonDataset.convert_pose_data_to_pose_map_and_vis(ds, None)
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "f.flo")
            with open(path, "wb") as f:
                f.write(np.float32(202021.25).tobytes() + np.int32(w).tobytes() + np.int32(h).tobytes() + flow.tobytes())
            ds.get_person_flow_path = lambda index: path
            ft, _ = td.TryonDataset.get_person_flow(ds, 0)
        ds.cloth_mask_threshold = 240     # the reference's default (--cloth_mask_threshold, tryon_dataset.py:74-79)
        cm_default = td.TryonDataset.get_input_cloth_mask(ds, im)
        ds.cloth_mask_threshold = 0.25
        cm_low = td.TryonDataset.get_input_cloth_mask(ds, im)
        data[f"{tag}:cloth_mask_240"] = cm_default.numpy()
        data[f"{tag}:cloth_mask_0.25"] = cm_low.numpy()
        data.update({f"{tag}:parse": parse, f"{tag}:image_u8": image, f"{tag}:keypoints": kp, f"{tag}:flow_payload": flow,
                     f"{tag}:image": im.numpy(), f"{tag}:im_head": head.numpy(), f"{tag}:im_cloth": cloth.numpy(),
                     f"{tag}:silhouette": sil.numpy(), f"{tag}:pose_map": pm.numpy(), f"{tag}:im_cocopose": vis.numpy(),
                     f"{tag}:pose_map_none": pm0.numpy(), f"{tag}:im_cocopose_none": vis0.numpy(), f"{tag}:flow": ft.numpy()})
        # the PIL-drawn squares themselves (what draw.rectangle paints), for the draw_into_map=1 mode of the kernel
        one = Image.new("L", (w, h))
        from PIL import ImageDraw
        squares = []
        for i in range(18):
            one_map = Image.new("L", (w, h))
            if kp[i, 0] > 1 and kp[i, 1] > 1:
                ImageDraw.Draw(one_map).rectangle((kp[i, 0] - 5, kp[i, 1] - 5, kp[i, 0] + 5, kp[i, 1] + 5), "white", "white")
            squares.append(np.array(one_map))
        data[f"{tag}:pil_squares"] = np.stack(squares)
        print(tag, "pose_map unique", np.unique(pm.numpy()), "im_cocopose unique", np.unique(vis.numpy()),
              "silhouette range", float(sil.min()), float(sil.max()))
    np.savez_compressed(os.path.join(out, "dataprep.npz"), **data)
    print("dataprep.npz", len(data), "arrays", os.path.getsize(os.path.join(out, "dataprep.npz")), "bytes")


def sams_hparams(**kw):
    """SamsModel options at a size the CPU finishes in seconds: 64 x 48 frames, 3 frames, features 8..32."""
    base = dict(n_frames_total=3, n_frames_now=None, person_inputs=["agnostic", "densepose", "flow"], cloth_inputs=["cloth"],
                encoder_input="flow", flow_warp=True, activation="relu", fine_height=64, fine_width=48,
                norm_G="spectralspadesyncbatch3x3", ngf_base=2, ngf_pow_outer=3, ngf_pow_inner=5, ngf_pow_step=1, num_middle=2,
                attention_middle_indices=[], attention_decoder_indices=[], init_type="xavier", init_variance=0.02,
                netD_subarch="n_layer", num_D=2, n_layers_D=4, ndf=8, norm_D="spectralinstance", gan_mode="hinge", lr=1e-4,
                lr_D=3e-4, no_ganFeat_loss=False, wt_l1=1.0, wt_vgg=1.0, wt_multiscale=1.0, wt_temporal=1.0)
    base.update(kw)
    return hp_namespace(**base)


SAMS_VARIANTS = {
    "base": dict(),
    "attn_gelu": dict(attention_middle_indices=["0"], attention_decoder_indices=["-1"], activation="gelu", gan_mode="ls",
                      norm_G="spectralspadebatch3x3"),
    "progressive": dict(n_frames_total=4, n_frames_now=2, flow_warp=False, gan_mode="original", norm_G="spadeinstance3x3",
                        norm_D="spectralbatch", no_ganFeat_loss=True, wt_l1=0.5, wt_vgg=2.0, wt_multiscale=0.25, wt_temporal=4.0),
}


def gen_sams(out):
    """SAMS-GAN: the reference's SamsModel, three training steps in Lightning's order (generator,
    multiscale discriminator, temporal discriminator) on procedural weights; every logged scalar, the generated frames,
    the buffers the steps mutate and the gradients of each step's own parameter set."""
    from models.sams_model import SamsModel

    for tag, kw in SAMS_VARIANTS.items():
        hp = sams_hparams(**kw)
        torch.manual_seed(0)
        model = SamsModel(hp)
        model.load_state_dict(procedural_state_dict(shapes_of(model.state_dict())))
        model.train()
        batch = synthetic_batch(2, "cpu", height=hp.fine_height, width=hp.fine_width, n_frames=hp.n_frames_total, smooth=True)
        data = {"state_keys": np.array(list(model.state_dict().keys())),
                "state_shapes": np.array([str(tuple(v.shape)) for v in model.state_dict().values()])}
        nets = {0: model.generator, 1: model.multiscale_discriminator, 2: model.temporal_discriminator}
        for idx in (0, 1, 2):
            # Lightning 0.9 with several optimizers: only the current optimizer's parameters require grad
            for p in model.parameters():
                p.requires_grad = False
            for p in nets[idx].parameters():
                p.requires_grad = True
            model.zero_grad()
            res = model.training_step(batch, 0, idx)
            res.minimize.backward()
            for k, v in res.logs.items():
                data[f"log{idx}:" + k] = np.float64(v.item())
            if idx == 0:
                data["frames_s4"] = strided(model.all_gen_frames, 4)
                data["frames_cs"] = checksums(model.all_gen_frames)
            own = {0: "generator", 1: "multiscale_discriminator", 2: "temporal_discriminator"}[idx]
            grads = {k: p.grad for k, p in nets[idx].named_parameters() if p.grad is not None}
            for k, g in grads.items():
                data[f"gcs{idx}:{own}.{k}"] = checksums(g)
                data[f"gs{idx}:{own}.{k}"] = g.detach().contiguous().reshape(-1)[::97].clone().numpy()
            # a few gradients in full: smallest tensors of the step plus the first conv
            for k in sorted(grads, key=lambda k: grads[k].numel())[:6] + [next(iter(grads))]:
                data[f"grad{idx}:{own}.{k}"] = grads[k].numpy()
            # buffers after this step (u, v, running statistics): checksums of all, a few in full
            for k, v in model.state_dict().items():
                if k.startswith("criterion_VGG") or not v.is_floating_point():
                    continue
                if k.endswith(("weight_u", "weight_v", "running_mean", "running_var")):
                    data[f"buf{idx}:{k}"] = checksums(v)
        for k, v in model.state_dict().items():
            if k.endswith("num_batches_tracked"):
                data["nbt:" + k] = np.int64(v.item())
        np.savez_compressed(os.path.join(out, f"sams_{tag}.npz"), **data)
        print(f"sams_{tag}.npz", {k: float(v) for k, v in data.items() if k.startswith("log")},
              os.path.getsize(os.path.join(out, f"sams_{tag}.npz")), "bytes")


def gen_sams_init(out):
    """BaseNetwork.init_weights (models/networks/base_network.py:42-77) under a fixed seed on the reference's own SAMS
    networks: every parameter after the call (the draw order covers the double initialisation of the sub-discriminators
    and the spectral-norm alias `module.weight` -> `weight_orig`)."""
    from models.networks import MultiscaleDiscriminator, NLayerDiscriminator
    from models.networks.sams.sams_generator import SamsGenerator

    data = {}
    for tag, kw in (("xavier", dict()), ("normal_batchD", dict(norm_D="spectralbatch", init_type="normal")),
                    ("kaiming_attn", dict(init_type="kaiming", attention_middle_indices=["0"], norm_G="spadeinstance3x3"))):
        hp = sams_hparams(**kw)
        torch.manual_seed(99)
        nets = {"G": SamsGenerator(hp), "Dm": MultiscaleDiscriminator(hp), "Dt": NLayerDiscriminator(hp, in_channels=15)}
        torch.manual_seed(1234)
        for name, net in nets.items():
            net.init_weights(hp.init_type, hp.init_variance)
            for k, p in net.named_parameters():
                data[f"{tag}:{name}.{k}"] = checksums(p)
            k0, p0 = next(iter(net.named_parameters()))
            data[f"{tag}:full:{name}.{k0}"] = p0.detach().numpy()
        data[f"{tag}:next_random"] = torch.rand(4).numpy()  # the generator's position after everything was drawn
    np.savez_compressed(os.path.join(out, "sams_init.npz"), **data)
    print("sams_init.npz", len(data), "arrays", os.path.getsize(os.path.join(out, "sams_init.npz")), "bytes")


if __name__ == "__main__":
    torch.set_num_threads(8)
    install_shim()
    which = sys.argv[1:] or ["ops", "warp", "unet"]
    if "ops" in which:
        gen_ops(HERE)
    if "warp" in which:
        gen_warp(HERE)
    if "unet" in which:
        gen_unet(HERE)
    if "nframes" in which or not sys.argv[1:]:
        gen_unet_nframes(HERE)
    if "init" in which or not sys.argv[1:]:
        gen_init(HERE)
    if "dataprep" in which or not sys.argv[1:]:
        gen_dataprep(HERE)
    if "sams" in which or not sys.argv[1:]:
        gen_sams(HERE)
    if "sams_init" in which or not sys.argv[1:]:
        gen_sams_init(HERE)
