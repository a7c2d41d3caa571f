"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

numpy restatement of the reference's dataset-side tensor preparation and of the image wire
format between the two stages.  Pinned by tests/golden/dataprep.npz, which tests/golden/make_golden.py produces by
calling the reference's own TryonDataset methods (datasets/tryon_dataset.py) on synthetic parse maps / keypoints /
images; tests/test_oracle_golden.py checks every function here against it, bit for bit.

Third-party arithmetic restated here because the reference delegates to it:
  * torchvision.transforms.ToTensor / Normalize (pinned torchvision 0.7.0, absent offline): byte/255, (t-mean)/std in fp32;
  * Pillow Image.resize(BILINEAR) (installed Pillow 12.2; src/libImaging/Resample.c): separable, coefficient tables in
    fp64, 22-bit fixed-point accumulation on 8-bit images, horizontal pass first;
  * Pillow ImageDraw.rectangle: float corners truncated with (int), both corners inclusive;
  * flownet2 `flow_utils.readFlow` (Middlebury .flo; submodule empty here => the FILE PARSE is UNPINNED, the
    normalisation that follows it is pinned).
"""
import numpy as np

# LIP parsing labels (datasets/tryon_dataset.py:21-41)
LIP = dict(BACKGROUND=0, HAT=1, HAIR=2, GLOVE=3, SUNGLASSES=4, UPPER_CLOTHES=5, DRESS=6, COAT=7, SOCKS=8, PANTS=9,
           JUMPSUITS=10, SCARF=11, SKIRT=12, FACE=13, LEFT_ARM=14, RIGHT_ARM=15, LEFT_LEG=16, RIGHT_LEG=17, LEFT_SHOE=18,
           RIGHT_SHOE=19)
# get_person_head (tryon_dataset.py:323-345): the labels summed into the "head" mask
HEAD_LABELS = ("HAT", "HAIR", "SUNGLASSES", "FACE", "SOCKS", "PANTS", "SCARF", "SKIRT", "LEFT_LEG", "RIGHT_LEG",
               "LEFT_SHOE", "RIGHT_SHOE")
# segment_cloths_from_image (datasets/util.py:6-22)
CLOTH_LABELS = ("UPPER_CLOTHES", "DRESS", "COAT")


def label_bits(names):
    bits = 0
    for n in names:
        bits |= 1 << LIP[n]
    return bits


def u8_to_normed(img_u8):
    """transforms.ToTensor + Normalize(0.5, 0.5) (tryon_dataset.py:109-118): (H, W[, C]) uint8 -> (C, H, W) fp32."""
    a = np.asarray(img_u8)
    if a.ndim == 2:
        a = a[:, :, None]
    t = a.transpose(2, 0, 1).astype(np.float32) / np.float32(255)
    return ((t - np.float32(0.5)) / np.float32(0.5)).astype(np.float32)


def quantise_u8(t):
    """visualization.py:73-77: (C, H, W) float -> uint8 (H, W, C) / (H, W): trunc(clamp((t + 1) * 0.5 * 255))."""
    a = (np.asarray(t, dtype=np.float32) + np.float32(1)) * np.float32(0.5) * np.float32(255)
    a = np.clip(a, 0, 255).astype(np.uint8)
    return a[0] if a.shape[0] == 1 else a.transpose(1, 2, 0)


def head_and_cloth(image, parse):
    """get_person_head (tryon_dataset.py:323-345) and segment_cloths_from_image (datasets/util.py:6-22).
    image (3, H, W) fp32, parse (H, W) uint8 labels -> im_head, im_cloth."""
    mh = np.isin(parse, [LIP[n] for n in HEAD_LABELS]).astype(np.float32)
    mc = np.isin(parse, [LIP[n] for n in CLOTH_LABELS]).astype(np.float32)
    return image * mh - (1 - mh), image * mc + (1 - mc)


def cloth_mask(cloth, threshold=240):
    """get_input_cloth_mask (tryon_dataset.py:168-175): where(cloth >= threshold, 0, 1)[0:1].  (The reference's default
    threshold 240 is compared with the NORMALISED cloth, so its mask is all ones.)"""
    c = np.asarray(cloth, np.float32)
    return np.where(c >= np.float32(threshold), np.float32(0), np.float32(1))[0:1].astype(np.float32)


def _pil_coeffs(in_size, out_size):
    """Pillow precompute_coeffs + normalize_coeffs_8bpc for the BILINEAR filter (support 1.0)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ss = 1.0 / filterscale
    bounds, coeffs = [], []
    for o in range(out_size):
        center = (o + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [max(0.0, 1.0 - abs((x + xmin - center + 0.5) * ss)) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        k = [int(v * (1 << 22) - 0.5) if v < 0 else int(v * (1 << 22) + 0.5) for v in w]
        bounds.append((xmin, xmax))
        coeffs.append(k)
    return bounds, coeffs


def _pil_pass(img, out_size, axis):
    """One separable 8-bit resampling pass along `axis` (1 = horizontal)."""
    img = np.asarray(img, dtype=np.uint8)
    a = img if axis == 1 else img.T
    bounds, coeffs = _pil_coeffs(a.shape[1], out_size)
    out = np.empty((a.shape[0], out_size), np.uint8)
    for o, ((xmin, xmax), k) in enumerate(zip(bounds, coeffs)):
        acc = np.full(a.shape[0], 1 << 21, np.int64)
        for x in range(xmax):
            acc += a[:, xmin + x].astype(np.int64) * k[x]
        out[:, o] = np.clip(acc >> 22, 0, 255).astype(np.uint8)
    return out if axis == 1 else out.T


def pil_resize_bilinear(img_u8, out_w, out_h):
    """PIL.Image.resize((out_w, out_h), Image.BILINEAR) on an 8-bit single-channel image: horizontal pass, then vertical."""
    a = np.asarray(img_u8, dtype=np.uint8)
    if a.shape[1] != out_w:
        a = _pil_pass(a, out_w, 1)
    if a.shape[0] != out_h:
        a = _pil_pass(a, out_h, 0)
    return a


def silhouette(parse, factor=16):
    """get_person_body_silhouette (tryon_dataset.py:347-369): (parse > 0) * 255 -> down /16 -> up -> normed (1, H, W)."""
    h, w = parse.shape
    shape = ((parse > 0).astype(np.float32) * 255).astype(np.uint8)
    small = pil_resize_bilinear(shape, w // factor, h // factor)
    return u8_to_normed(pil_resize_bilinear(small, w, h))


def pose_map(pose_data, height=256, width=192, radius=5, draw_into_map=False):
    """convert_pose_data_to_pose_map_and_vis (tryon_dataset.py:388-448).  pose_data (P, 3) float64 or None.
    Returns (pose_map (P, H, W), im_cocopose (1, H, W)), values in {-1, +1}.
    draw_into_map=False is the reference AS WRITTEN: each plane is converted to a tensor (:417-424) BEFORE the square is
    drawn on its PIL image (:426-434), so the planes stay at -1 and only the visual carries the squares."""
    p = pose_data.shape[0] if pose_data is not None else 18
    vis = -np.ones((1, height, width), np.float32)
    if pose_data is None:  # no detected person: the planes keep the torch.zeros they were allocated with (:403-405)
        return np.zeros((p, height, width), np.float32), vis
    maps = -np.ones((p, height, width), np.float32)
    if pose_data is not None:
        for i in range(p):
            x, y = float(pose_data[i, 0]), float(pose_data[i, 1])
            if x > 1 and y > 1:
                x0, x1, y0, y1 = int(x - radius), int(x + radius), int(y - radius), int(y + radius)
                xs, ys = slice(max(x0, 0), max(x1 + 1, 0)), slice(max(y0, 0), max(y1 + 1, 0))
                vis[0, ys, xs] = 1.0
                if draw_into_map:
                    maps[i, ys, xs] = 1.0
    return maps, vis


def read_flo(raw):
    """UNPINNED file parse (flownet2 flow_utils.readFlow, Middlebury format): float32 magic 202021.25, int32 width,
    int32 height, then height*width*2 float32 (u, v interleaved).  Returns (H, W, 2)."""
    magic = np.frombuffer(raw, np.float32, 1, 0)[0]
    if magic != np.float32(202021.25):
        raise ValueError("not a .flo file")
    w, h = (int(v) for v in np.frombuffer(raw, np.int32, 2, 4))
    return np.frombuffer(raw, np.float32, h * w * 2, 12).reshape(h, w, 2).copy()


def flow_tensor(flow_hw2):
    """get_person_flow (tryon_dataset.py:283-289): (H, W, 2) -> permute(2, 0, 1) -> Normalize((.5, .5), (.5, .5))."""
    t = np.asarray(flow_hw2, np.float32).transpose(2, 0, 1)
    return ((t - np.float32(0.5)) / np.float32(0.5)).astype(np.float32)
