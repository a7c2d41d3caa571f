"""GPU: dataset-side tensor preparation (csrc/dataprep.hip) and the inter-stage image wire format, bit-exact against
the oracle (oracle/dataprep_oracle.py, pinned to the reference's TryonDataset methods by tests/golden/dataprep.npz) and
against the golden itself.  Reference: datasets/tryon_dataset.py:109-121,226-229,272-298,323-448, datasets/util.py:6-22,
visualization.py:60-88, datasets/vvt_dataset.py:139-150."""
import os

import numpy as np
import pytest
import torch

from helpers import load_golden, make_namespace, oracle
from oracle import dataprep_oracle as dpo

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["full", "small"])
def test_dataprep_kernels_bit_exact_vs_reference_golden(cuda, tag):
    from shineon_virtual_tryon_amd import dataprep

    g = load_golden("dataprep.npz")
    parse, image_u8, kp, payload = (g[f"{tag}:{k}"] for k in ("parse", "image_u8", "keypoints", "flow_payload"))
    h, w = parse.shape
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)  # noqa: E731
    # a batch of two: the golden sample and a second one built by flipping it (exercises the batch strides)
    parse2, image2 = np.stack([parse, parse[::-1, ::-1]]), np.stack([image_u8, image_u8[::-1, ::-1]])
    kp2 = np.stack([kp, np.full_like(kp, np.nan)])          # second sample: no detected person
    im = dataprep.images_to_normed(dev(image2))
    np.testing.assert_array_equal(im[0].cpu().numpy(), g[f"{tag}:image"])
    rep = dataprep.person_representation(dev(parse2), im)
    np.testing.assert_array_equal(rep["im_head"][0].cpu().numpy(), g[f"{tag}:im_head"])
    np.testing.assert_array_equal(rep["im_cloth"][0].cpu().numpy(), g[f"{tag}:im_cloth"])
    np.testing.assert_array_equal(rep["silhouette"][0].cpu().numpy(), g[f"{tag}:silhouette"])
    assert rep["agnostic"].shape == (2, 4, h, w)            # [silhouette | im_head], tryon_dataset.py:226-229
    # second sample against the oracle
    im1 = dpo.u8_to_normed(image2[1])
    head1, cloth1 = dpo.head_and_cloth(im1, parse2[1])
    np.testing.assert_array_equal(rep["agnostic"][1].cpu().numpy(), np.concatenate([dpo.silhouette(parse2[1]), head1]))
    np.testing.assert_array_equal(rep["im_cloth"][1].cpu().numpy(), cloth1)
    # pose maps: the reference as written (planes stay -1), the no-person sample (zeros), and the painted mode vs PIL
    maps, vis = dataprep.pose_maps(dev(kp2), h, w, radius=5)
    np.testing.assert_array_equal(maps[0].cpu().numpy(), g[f"{tag}:pose_map"])
    np.testing.assert_array_equal(vis[0].cpu().numpy(), g[f"{tag}:im_cocopose"])
    np.testing.assert_array_equal(maps[1].cpu().numpy(), g[f"{tag}:pose_map_none"])
    np.testing.assert_array_equal(vis[1].cpu().numpy(), g[f"{tag}:im_cocopose_none"])
    painted, vis_p = dataprep.pose_maps(dev(kp2), h, w, radius=5, draw_into_map=True)
    np.testing.assert_array_equal(painted[0].cpu().numpy(), g[f"{tag}:pil_squares"].astype(np.float32) / 255 * 2 - 1)
    np.testing.assert_array_equal(vis_p.cpu().numpy(), vis.cpu().numpy())
    # flow
    fl = dataprep.flow_from_payload(dev(np.stack([payload, -payload])))
    np.testing.assert_array_equal(fl[0].cpu().numpy(), g[f"{tag}:flow"])
    np.testing.assert_array_equal(fl[1].cpu().numpy(), dpo.flow_tensor(-payload))
    assert np.array_equal(dataprep.read_flo(np.float32(202021.25).tobytes() + np.int32(w).tobytes() + np.int32(h).tobytes()
                                            + payload.tobytes()).numpy(), payload)


def test_silhouette_random_masks_and_sizes_vs_pil(cuda):
    """The Pillow resampling restatement on sizes whose /16 grid is ragged, against PIL itself on the box."""
    from PIL import Image

    from shineon_virtual_tryon_amd import dataprep

    rng = np.random.default_rng(11)
    for h, w in ((256, 192), (200, 150), (64, 48), (97, 131)):
        parse = (rng.random((3, h, w)) < 0.5).astype(np.uint8) * rng.integers(1, 20, size=(3, h, w)).astype(np.uint8)
        image = torch.zeros(3, 3, h, w, device=cuda)
        rep = dataprep.person_representation(torch.from_numpy(parse).to(cuda), image)
        for i in range(3):
            shape = Image.fromarray(((parse[i] > 0).astype(np.float32) * 255).astype(np.uint8))
            ref = shape.resize((w // 16, h // 16), Image.BILINEAR).resize((w, h), Image.BILINEAR)
            np.testing.assert_array_equal(rep["silhouette"][i].cpu().numpy(), dpo.u8_to_normed(np.array(ref)),
                                          err_msg=f"{h}x{w} sample {i}")


def test_quantise_and_dequantise_kernels(cuda):
    from shineon_virtual_tryon_amd import dataprep, ops

    g = torch.Generator().manual_seed(4)
    t = (torch.rand(3, 3, 40, 24, generator=g) * 2.6 - 1.3)
    t[0, 0, 0, :6] = torch.tensor([-1.0, 1.0, 0.0, -1.0000001, 0.99999994, 254.5 / 127.5 - 1])
    ref = np.stack([dpo.quantise_u8(x.numpy()) for x in t])
    np.testing.assert_array_equal(dataprep.quantise_images(t.to(cuda)).cpu().numpy(), ref)
    rows = ops.to_rows(t.to(cuda), cpad=4)[:, :3]          # NHWC-pitch source (how p_tryon leaves the model)
    assert not rows.is_contiguous()
    np.testing.assert_array_equal(dataprep.quantise_images(rows).cpu().numpy(), ref)
    mask = t[:, :1]
    np.testing.assert_array_equal(dataprep.quantise_images(mask.to(cuda).contiguous()).cpu().numpy()[..., 0],
                                  np.stack([dpo.quantise_u8(x.numpy()) for x in mask]))
    u8 = torch.from_numpy(ref)
    back = dataprep.images_to_normed(u8.to(cuda)).cpu().numpy()
    np.testing.assert_array_equal(back, np.stack([dpo.u8_to_normed(a) for a in ref]))
    every = torch.arange(256, dtype=torch.uint8).reshape(1, 16, 16, 1)
    np.testing.assert_array_equal(dataprep.images_to_normed(every.to(cuda)).cpu().numpy()[0],
                                  dpo.u8_to_normed(every[0].numpy()))


def test_warp_test_step_to_disk_to_tryon_training_step(cuda, tmp_path):
    """f1 end to end: WarpModel.test_step writes warp-cloth/ PNGs -> io_png.load_images reads them back as the try-on
    stage's `cloth` input -> UnetMaskModel.training_step.  The bytes on disk are the reference's quantisation of the
    warped cloth, the tensor read back is ToTensor+Normalize of those bytes, and the try-on losses equal the oracle's on
    that same input (the reference's hand-off: warp_model.py:143-149 -> vvt_dataset.py:139-150)."""
    from PIL import Image

    from oracle.procedural import procedural_state_dict, shapes_of
    from shineon_virtual_tryon_amd import io_png
    from shineon_virtual_tryon_amd.data import synthetic_batch
    from shineon_virtual_tryon_amd.unet_mask_model import UnetMaskModel
    from shineon_virtual_tryon_amd.warp_model import WarpModel

    hp = make_namespace(is_train=False, person_inputs=["agnostic", "cocopose"], checkpoint="ckpt/w.ckpt", name="gmm",
                        result_dir=str(tmp_path), datamode="test")
    warp = WarpModel(hp)
    warp.load_state_dict(procedural_state_dict(shapes_of(warp.state_dict())))
    warp = warp.to(cuda).eval()
    warp.override_hparams(hp)
    batch_cpu = synthetic_batch(2, "cpu", smooth=True)
    batch = {k: (v.to(cuda) if isinstance(v, torch.Tensor) else v) for k, v in batch_cpu.items()}
    names = batch["cloth_name"][0]
    with torch.no_grad():
        warp.test_step(batch, 0)
    root = os.path.join(str(tmp_path), "gmm", "w.ckpt", "test")
    paths = [io_png.find_warp_cloth(root, n, "SyntheticDataset") for n in names]
    on_disk = np.stack([np.array(Image.open(p)) for p in paths])
    expected = np.stack([dpo.quantise_u8(x.cpu().numpy()) for x in warp.warped_cloth])
    np.testing.assert_array_equal(on_disk, expected)
    cloth = io_png.load_images(paths, cuda)
    np.testing.assert_array_equal(cloth.cpu().numpy(), np.stack([dpo.u8_to_normed(a) for a in on_disk]))
    assert float((cloth - warp.warped_cloth).abs().max()) <= 2.0 / 255 + 1e-6

    unet = UnetMaskModel(make_namespace(self_attn=True, activation="gelu"))
    usd = procedural_state_dict(shapes_of(unet.state_dict()))
    unet.load_state_dict(usd)
    unet = unet.to(cuda).train()
    b2 = dict(batch)
    b2["cloth"] = cloth
    res = unet.training_step(b2, 0)
    b2c = dict(batch_cpu)
    b2c["cloth"] = cloth.cpu()
    uhp = dict(n_frames_total=1, person_inputs=["agnostic", "densepose"], cloth_inputs=["cloth"], self_attn=True, num_attn=2,
               activation="gelu", flow_warp=False)
    with torch.no_grad():
        ref = oracle.unet_mask_losses(usd, b2c, uhp)
    for k in ("loss/G", "loss/G/l1", "loss/G/vgg", "loss/G/tryon_mask_l1"):
        r = float(ref[k])
        assert abs(float(res.logs[k]) - r) <= 2e-5 + 2e-5 * abs(r), (k, float(res.logs[k]), r)
    assert float((unet.p_tryons[0].cpu() - ref["p_tryons"]).abs().max()) < 1e-4


def test_build_batch_equals_the_reference_getitem_tensors(cuda):
    """dataprep.build_batch - the tensor part of TryonDataset.__getitem__ for a whole batch on the GPU - against the
    reference's own outputs (tests/golden/dataprep.npz), bit for bit, including the cloth mask (all ones with the reference's
    default threshold) and the zeros the reference substitutes for missing optional files."""
    from shineon_virtual_tryon_amd import dataprep

    g = load_golden("dataprep.npz")
    tag = "full"
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)  # noqa: E731
    parse, image_u8, kp, payload = (g[f"{tag}:{k}"] for k in ("parse", "image_u8", "keypoints", "flow_payload"))
    raw = {"image_u8": dev(image_u8[None]), "cloth_u8": dev(image_u8[None]), "parse_u8": dev(parse[None]),
           "keypoints": dev(kp[None]), "flow_payload": dev(payload[None])}
    b = dataprep.build_batch(raw)
    expect = {"image": "image", "cloth": "image", "im_head": "im_head", "im_cloth": "im_cloth", "silhouette": "silhouette",
              "cocopose": "pose_map", "im_cocopose": "im_cocopose", "flow": "flow", "cloth_mask": "cloth_mask_240"}
    for key, gk in expect.items():
        np.testing.assert_array_equal(b[key][0].cpu().numpy(), g[f"{tag}:{gk}"], err_msg=key)
    np.testing.assert_array_equal(b["agnostic"][0].cpu().numpy(), np.concatenate([g[f"{tag}:silhouette"], g[f"{tag}:im_head"]]))
    assert float(b["prev_image"].abs().max()) == 0.0 and float(b["densepose"].abs().max()) == 0.0   # missing files -> zeros
    low = dataprep.cloth_mask(b["cloth"], 0.25)
    np.testing.assert_array_equal(low[0].cpu().numpy(), g[f"{tag}:cloth_mask_0.25"])
    # the batch feeds the models: shapes / keys of the dict the reference's collate would hand to training_step
    assert b["agnostic"].shape == (1, 4, 256, 192) and b["cocopose"].shape == (1, 18, 256, 192) and b["cloth_mask"].shape == (1, 1, 256, 192)
