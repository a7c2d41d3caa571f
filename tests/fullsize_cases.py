"""The full-size parity cases: inputs, hyper-parameters and procedural weights, shared by the fixture generator
(tests/golden/make_golden_fullsize.py, build container, evaluates the oracle) and the `-m gpu` tests (GPU box, evaluate
the HIP path and compare with the committed fixtures).  Nothing here touches the oracle."""
import torch

from helpers import make_namespace
from oracle.procedural import procedural_state_dict, shapes_of

WHP = dict(person_inputs=["agnostic", "cocopose"], cloth_inputs=["cloth"])
UHP = dict(n_frames_total=1, person_inputs=["agnostic", "densepose"], cloth_inputs=["cloth"], self_attn=True, num_attn=2,
           activation="gelu", flow_warp=False, pen_flow_mask=1.0)
C5HP = dict(UHP, n_frames_total=5, flow_warp=True)
UNET_LOG_KEYS = ("loss/G", "loss/G/l1", "loss/G/vgg", "loss/G/tryon_mask_l1", "loss/G/flow_mask_l1")


def smooth_batch(bs, **kw):
    from shineon_virtual_tryon_amd.data import synthetic_batch

    return synthetic_batch(bs, "cpu", smooth=True, **kw)


def to_device(batch, dev):
    return {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}


def build(cls, device=None, **hp):
    """Model with procedural weights (seed 420, key-addressed) -> (model, state_dict on the CPU)."""
    model = cls(make_namespace(**hp))
    sd = procedural_state_dict(shapes_of(model.state_dict()))
    model.load_state_dict(sd, strict=True)
    if device is not None:
        model = model.to(device)
    return model.train(), sd


def build_warp(device=None):
    from shineon_virtual_tryon_amd.warp_model import WarpModel

    return build(WarpModel, device, person_inputs=["agnostic", "cocopose"])


def build_unet(device=None):
    from shineon_virtual_tryon_amd.unet_mask_model import UnetMaskModel

    return build(UnetMaskModel, device, self_attn=True, activation="gelu")


def build_c5(device=None):
    from shineon_virtual_tryon_amd.unet_mask_model import UnetMaskModel

    return build(UnetMaskModel, device, n_frames_total=5, flow_warp=True, activation="gelu", self_attn=True)


def flatten_frames(batch):
    return {k: (v.reshape(v.shape[0], -1, *v.shape[3:]) if isinstance(v, torch.Tensor) and v.dim() == 5 else v)
            for k, v in batch.items()}


def handoff_cloth(warped_oracle):
    """The try-on stage's cloth input of the chained cases: the ORACLE's warped cloth rounded to fp16 (exactly
    representable in fp32, 1/2 the fixture bytes).  The GPU's own warped cloth is checked against the oracle's separately;
    feeding both sides this one tensor keeps the try-on comparison free of the warp stage's 1e-5 differences."""
    return warped_oracle.detach().to(torch.float16).to(torch.float32)


# ---- SAMS ------------------------------------------------------------------------------------------
def sams_full_hparams(**kw):
    import bench

    return bench.sams_hparams(**kw)


def sams_full_three_steps_case():
    """bench.py --config sams networks (reference defaults, 184.8 M generator parameters), 256x192, bs=1, n_frames_total=3."""
    from shineon_virtual_tryon_amd.data import synthetic_batch
    from shineon_virtual_tryon_amd.sams_model import SamsModel

    hp = sams_full_hparams(n_frames_total=3)
    model = SamsModel(hp)
    sd = procedural_state_dict(shapes_of(model.state_dict()))
    batch = synthetic_batch(1, "cpu", n_frames=hp.n_frames_total, smooth=True)
    return hp, model, sd, batch


def sams_full_generator_case():
    """The reference-default generator at bs=4, 256x192, four previous frames: inputs of one forward + backward pass."""
    from shineon_virtual_tryon_amd.data import synthetic_batch
    from shineon_virtual_tryon_amd.networks.sams.sams_generator import SamsGenerator

    hp = sams_full_hparams()
    gen = SamsGenerator(hp)
    sd = procedural_state_dict({"generator." + k: v for k, v in shapes_of(gen.state_dict()).items()})
    torch.manual_seed(12)
    b, n, h, w = 4, hp.n_frames_total, hp.fine_height, hp.fine_width
    batch = synthetic_batch(b, "cpu", n_frames=n, smooth=True)
    prev_frames = batch["image"][:, :n - 1].contiguous()
    prev_maps = batch["flow"][:, :n - 1].contiguous()
    maps = {k: batch[k][:, -1].contiguous() for k in ("agnostic", "densepose", "flow", "cloth")}
    gout = torch.randn(b, 4, h, w) / (h * w)
    return hp, gen, sd, prev_frames, prev_maps, maps, gout


def sams_small_case(tag):
    import sams_helpers as sh
    from shineon_virtual_tryon_amd.data import synthetic_batch

    g = sh.load_golden(tag)
    sd = procedural_state_dict(sh.golden_shapes(g))
    hp = sh.sams_hparams(**sh.SAMS_VARIANTS[tag])
    batch = synthetic_batch(2, "cpu", height=hp.fine_height, width=hp.fine_width, n_frames=hp.n_frames_total, smooth=True)
    return g, sd, hp, batch
