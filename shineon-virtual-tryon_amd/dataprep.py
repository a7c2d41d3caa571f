"""Dataset-side tensor preparation on the GPU (reference: datasets/tryon_dataset.py:109-121,226-229,272-298,323-448,
datasets/util.py:6-22 — per-sample PIL / numpy / torchvision work the reference itself calls "very expensive").

The raw inputs of a batch are uploaded as they are stored on disk — uint8 images (HWC), the uint8 LIP parse map, the
18 x 3 fp64 keypoints, the .flo payload — and every derived tensor of the batch dict is produced by csrc/dataprep.hip,
bit-exact with the reference's CPU arithmetic (tests/test_dataprep_gpu.py against golden vectors produced by the
reference's own TryonDataset methods).  Host code here only allocates outputs and launches.
"""
import torch

from ._lib import check, lib
from .ops import _require_cuda, _stream

# LIP parsing labels (datasets/tryon_dataset.py:21-41) and the two label sets the reference isolates
LIP = dict(BACKGROUND=0, HAT=1, HAIR=2, GLOVE=3, SUNGLASSES=4, UPPER_CLOTHES=5, DRESS=6, COAT=7, SOCKS=8, PANTS=9,
           JUMPSUITS=10, SCARF=11, SKIRT=12, FACE=13, LEFT_ARM=14, RIGHT_ARM=15, LEFT_LEG=16, RIGHT_LEG=17, LEFT_SHOE=18,
           RIGHT_SHOE=19)
HEAD_LABELS = ("HAT", "HAIR", "SUNGLASSES", "FACE", "SOCKS", "PANTS", "SCARF", "SKIRT", "LEFT_LEG", "RIGHT_LEG",
               "LEFT_SHOE", "RIGHT_SHOE")                     # get_person_head, tryon_dataset.py:326-341
CLOTH_LABELS = ("UPPER_CLOTHES", "DRESS", "COAT")             # segment_cloths_from_image, datasets/util.py:13-17


def _bits(names):
    b = 0
    for n in names:
        b |= 1 << LIP[n]
    return b


HEAD_BITS, CLOTH_BITS = _bits(HEAD_LABELS), _bits(CLOTH_LABELS)


def _u8(t, what):
    _require_cuda(t)
    if t.dtype != torch.uint8:
        raise TypeError(f"{what} must be uint8, got {t.dtype}")
    return t.contiguous()


def images_to_normed(images_u8, channels=None):
    """ToTensor + Normalize(0.5, 0.5): (N, H, W, Cs) or (N, H, W) uint8 -> (N, C, H, W) fp32 in [-1, 1]."""
    u = _u8(images_u8, "images")
    if u.dim() == 3:
        u = u.unsqueeze(-1)
    n, h, w, cs = u.shape
    c = cs if channels is None else channels
    out = torch.empty((n, c, h, w), dtype=torch.float32, device=u.device)
    check(lib().so_u8_to_normed(u.data_ptr(), cs, out.data_ptr(), n, c, h * w, _stream()), "u8_to_normed")
    return out


def quantise_images(t):
    """The PNG wire format's quantisation (visualization.py:73-77): (N, C, H, W) float -> (N, H, W, C) uint8."""
    from .ops import _is_rows, _ld

    _require_cuda(t)
    n, c, h, w = t.shape
    out = torch.empty((n, h, w, c), dtype=torch.uint8, device=t.device)
    if t.is_contiguous():
        check(lib().so_quantize_u8(t.data_ptr(), 0, 1, out.data_ptr(), n, c, h * w, _stream()), "quantize_u8")
    elif _is_rows(t):
        check(lib().so_quantize_u8(t.data_ptr(), _ld(t), 0, out.data_ptr(), n, c, h * w, _stream()), "quantize_u8")
    else:
        t = t.contiguous()
        check(lib().so_quantize_u8(t.data_ptr(), 0, 1, out.data_ptr(), n, c, h * w, _stream()), "quantize_u8")
    return out


def pose_maps(keypoints, height, width, radius=5, draw_into_map=False):
    """(N, P, 3) fp64 keypoints (NaN rows = no detected person) -> cocopose (N, P, H, W), im_cocopose (N, 1, H, W).
    draw_into_map=False is the reference as written (planes stay -1, see csrc/dataprep.hip); True paints the squares."""
    _require_cuda(keypoints)
    kp = keypoints.to(torch.float64).contiguous()
    n, p, _ = kp.shape
    maps = torch.empty((n, p, height, width), dtype=torch.float32, device=kp.device)
    vis = torch.empty((n, 1, height, width), dtype=torch.float32, device=kp.device)
    check(lib().so_pose_map(kp.data_ptr(), maps.data_ptr(), vis.data_ptr(), n, p, height, width, int(radius),
                            int(bool(draw_into_map)), _stream()), "pose_map")
    return maps, vis


def person_representation(parse_u8, image, factor=16):
    """parse (N, H, W) uint8 LIP labels + image (N, 3, H, W) normed -> dict(silhouette, im_head, im_cloth, agnostic):
    agnostic = cat([silhouette, im_head]) is written in place (tryon_dataset.py:226-229)."""
    parse = _u8(parse_u8, "parse")
    _require_cuda(image)
    image = image.contiguous()
    n, h, w = parse.shape
    dev = parse.device
    L = lib()
    agnostic = torch.empty((n, 4, h, w), dtype=torch.float32, device=dev)
    im_cloth = torch.empty((n, 3, h, w), dtype=torch.float32, device=dev)
    shape = torch.empty((n, h, w), dtype=torch.uint8, device=dev)
    hw = h * w
    head_ptr = agnostic.data_ptr() + 4 * hw  # plane 1 of every sample
    check(L.so_parse_compose(parse.data_ptr(), image.data_ptr(), head_ptr, 4 * hw, im_cloth.data_ptr(), 3 * hw,
                             shape.data_ptr(), HEAD_BITS, CLOTH_BITS, n, hw, _stream()), "parse_compose")
    ws = torch.empty(max(1, L.so_silhouette_ws_bytes(n, h, w, factor)), dtype=torch.uint8, device=dev)
    check(L.so_silhouette(shape.data_ptr(), agnostic.data_ptr(), 4 * hw, ws.data_ptr(), n, h, w, factor, _stream()),
          "silhouette")
    return {"agnostic": agnostic, "silhouette": agnostic[:, 0:1], "im_head": agnostic[:, 1:4], "im_cloth": im_cloth}


def flow_from_payload(payload):
    """(N, H, W, 2) fp32 .flo payload -> (N, 2, H, W) normalised flow (tryon_dataset.py:283-289)."""
    _require_cuda(payload)
    p = payload.to(torch.float32).contiguous()
    n, h, w, _ = p.shape
    out = torch.empty((n, 2, h, w), dtype=torch.float32, device=p.device)
    check(lib().so_flow_decode(p.data_ptr(), out.data_ptr(), n, h * w, _stream()), "flow_decode")
    return out


def cloth_mask(cloth, threshold=240.0):
    """get_input_cloth_mask (tryon_dataset.py:168-175): (cloth[:, 0] >= threshold) ? 0 : 1 -> (N, 1, H, W).  The reference
    applies its default threshold 240 (--cloth_mask_threshold, a 0-255 value) to the NORMALISED cloth in [-1, 1], so with the
    default the mask is all ones; that behaviour is reproduced as is."""
    _require_cuda(cloth)
    c = cloth.contiguous()
    n, ch, h, w = c.shape
    out = torch.empty((n, 1, h, w), dtype=torch.float32, device=c.device)
    check(lib().so_threshold_mask(c.data_ptr(), ch, float(threshold), out.data_ptr(), n, h * w, _stream()), "threshold_mask")
    return out


def build_batch(raw, radius=5, cloth_mask_threshold=240.0, draw_pose_into_map=False):
    """The tensor part of TryonDataset.__getitem__ (tryon_dataset.py:481-537 = get_cloth_representation :209-250 +
    get_input_cloth / get_input_cloth_mask :156-184 + get_person_flow :272-298) for a whole batch, on the GPU.

    raw: device tensors as stored on disk -
      image_u8 (N, H, W, 3), cloth_u8 (N, H, W, 3), parse_u8 (N, H, W) LIP labels, keypoints (N, 18, 3) fp64 (NaN = no person);
      optional prev_image_u8, densepose_u8 (N, H, W, 3), flow_payload (N, H, W, 2) fp32, grid_u8 (N, H, W, 3).
    Returns the batch dict entries the models read (image, prev_image, cloth, cloth_mask, im_cloth, silhouette, im_head,
    agnostic, cocopose, im_cocopose, densepose, [flow], [grid_vis]); missing optional inputs give the zeros the reference
    substitutes for a missing file (:262-266, :293-294, :311-312)."""
    image = images_to_normed(raw["image_u8"])
    n, _, h, w = image.shape
    dev = image.device
    out = {"image": image}
    out["prev_image"] = images_to_normed(raw["prev_image_u8"]) if raw.get("prev_image_u8") is not None else \
        _zeros((n, 3, h, w), dev)
    out["cloth"] = images_to_normed(raw["cloth_u8"])
    out["cloth_mask"] = cloth_mask(out["cloth"], cloth_mask_threshold)
    rep = person_representation(raw["parse_u8"], image)
    out.update(silhouette=rep["silhouette"], im_head=rep["im_head"], im_cloth=rep["im_cloth"], agnostic=rep["agnostic"])
    out["cocopose"], out["im_cocopose"] = pose_maps(raw["keypoints"], h, w, radius, draw_pose_into_map)
    out["densepose"] = images_to_normed(raw["densepose_u8"]) if raw.get("densepose_u8") is not None else _zeros((n, 3, h, w), dev)
    if raw.get("flow_payload") is not None:
        out["flow"] = flow_from_payload(raw["flow_payload"])
    if raw.get("grid_u8") is not None:
        out["grid_vis"] = images_to_normed(raw["grid_u8"])
    return out


def _zeros(shape, device):
    from .ops import fill_

    return fill_(torch.empty(shape, dtype=torch.float32, device=device), 0.0)


def read_flo(raw):
    """Middlebury .flo container (flownet2 flow_utils.readFlow; upstream source absent => file parse unpinned):
    float32 magic 202021.25, int32 width, int32 height, H*W*2 float32.  Returns a (H, W, 2) CPU tensor."""
    import numpy as np

    if np.frombuffer(raw, np.float32, 1, 0)[0] != np.float32(202021.25):
        raise ValueError("not a .flo file (bad magic)")
    w, h = (int(v) for v in np.frombuffer(raw, np.int32, 2, 4))
    return torch.from_numpy(np.frombuffer(raw, np.float32, h * w * 2, 12).reshape(h, w, 2).copy())
