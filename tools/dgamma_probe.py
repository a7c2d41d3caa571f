"""Where does the error of the attention gamma gradient come from?  (ours 5.3e-2 from fp64, the fp32
reference 1.1e-2.)   d gamma = <dout, o>: a cancelling dot product of the gradient arriving at the attention output and the
attention result o = softmax(q k^T) v.  This probe runs UnetMaskModel (bs=4, attn+gelu) on the GPU and the oracle in fp32 and
fp64, captures (dout, o) of every attention module on all three, and splits each error into its two first-order parts

    d gamma_X - d gamma_64  ~=  <dout_X - dout_64, o_64>  +  <dout_64, o_X - o_64>

printed relative to |d gamma_64|, together with the element-wise distance of dout / o from fp64.  Usage (GPU box):
    python tools/dgamma_probe.py [--bs 4]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import make_namespace, oracle  # noqa: E402
from oracle.procedural import procedural_state_dict, shapes_of  # noqa: E402


def run_oracle(sd, batch, hp, dtype):
    caps = []
    orig = oracle.self_attention

    def wrapped(x, sdict, prefix):
        import torch.nn.functional as F

        b, c, h, w = x.shape
        n = h * w
        q = F.conv2d(x, sdict[prefix + ".query_conv.weight"], sdict[prefix + ".query_conv.bias"]).view(b, -1, n)
        k = F.conv2d(x, sdict[prefix + ".key_conv.weight"], sdict[prefix + ".key_conv.bias"]).view(b, -1, n)
        v = F.conv2d(x, sdict[prefix + ".value_conv.weight"], sdict[prefix + ".value_conv.bias"]).view(b, -1, n)
        attn = torch.softmax(torch.einsum("bdi,bdj->bij", q, k), dim=-1)
        out = torch.einsum("bcj,bij->bci", v, attn).reshape(b, c, h, w)
        res = sdict[prefix + ".gamma"] * out + x
        rec = {"prefix": prefix, "o": out.detach()}
        res.register_hook(lambda g, rec=rec: rec.__setitem__("dout", g.detach()))
        caps.append(rec)
        return res

    oracle.self_attention = wrapped
    try:
        params = {k: (v.to(dtype).clone().requires_grad_(k.startswith("unet.")) if v.is_floating_point() else v.clone())
                  for k, v in sd.items()}
        b = {k: (v.to(dtype) if isinstance(v, torch.Tensor) and v.is_floating_point() else v) for k, v in batch.items()}
        oracle.unet_mask_losses(params, b, hp)["loss/G"].backward()
    finally:
        oracle.self_attention = orig
    return {r["prefix"]: r for r in caps}, params


def run_gpu(sd, batch, dev):
    from shineon_virtual_tryon_amd import ops
    from shineon_virtual_tryon_amd.unet_mask_model import UnetMaskModel

    model = UnetMaskModel(make_namespace(self_attn=True, activation="gelu"))
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).train()
    model.configure_optimizers()  # plants the flat gradient slab: the fused q/k/v attention path runs, as in bench.py
    caps = []
    for cls in (ops._SelfAttentionQkvFn, ops._SelfAttentionFn):
        orig = cls.backward

        def make(orig, cls):
            def backward(ctx, dout):
                saved = ctx.saved_tensors
                o = saved[3] if cls is ops._SelfAttentionQkvFn else saved[8]
                x = saved[0]
                b, c, h, w = x.shape
                d = ops._dense_rows(dout)
                caps.append({"o": o.detach().view(b, h, w, c).permute(0, 3, 1, 2).cpu().double(),
                             "dout": d.detach().cpu().double().contiguous()})
                return orig(ctx, dout)
            return backward

        cls.backward = staticmethod(make(orig, cls))
    gb = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    res = model.training_step(gb, 0)
    res.minimize.backward()
    torch.cuda.synchronize()
    return caps, model


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", type=int, default=4)
    args = ap.parse_args()
    from shineon_virtual_tryon_amd.data import synthetic_batch
    from shineon_virtual_tryon_amd.unet_mask_model import UnetMaskModel

    dev = torch.device("cuda", 0)
    hp = dict(n_frames_total=1, person_inputs=["agnostic", "densepose"], cloth_inputs=["cloth"], self_attn=True, num_attn=2,
              activation="gelu", flow_warp=False, pen_flow_mask=1.0)
    sd = procedural_state_dict(shapes_of(UnetMaskModel(make_namespace(self_attn=True, activation="gelu")).state_dict()))
    batch = synthetic_batch(args.bs, "cpu", smooth=True)
    c32, p32 = run_oracle(sd, batch, hp, torch.float32)
    c64, p64 = run_oracle(sd, batch, hp, torch.float64)
    ours, model = run_gpu(sd, batch, dev)
    # oracle modules are visited outermost-first in forward; our backward visits them in reverse order of the forward
    names = list(c64.keys())
    ours = ours[::-1]
    assert len(ours) == len(names), (len(ours), names)
    grads = {n: p.grad.detach().cpu().double() for n, p in model.named_parameters() if n.endswith("gamma")}
    dot = lambda a, b: float((a.double() * b.double()).sum())  # noqa: E731
    rel = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max())  # noqa: E731
    for name, mine in zip(names, ours):
        r64, r32 = c64[name], c32[name]
        g64 = dot(r64["dout"], r64["o"])
        mass = float((r64["dout"] * r64["o"]).abs().sum())
        print(f"== {name}: d gamma (fp64) {g64:+.6e}, sum|terms| {mass:.3e} (cancellation x{mass / abs(g64):.0f}), "
              f"{r64['o'].numel()} terms")
        for tag, r in (("fp32 reference", r32), ("HIP", mine)):
            g = dot(r["dout"], r["o"])
            via_dout = dot(r["dout"].double() - r64["dout"], r64["o"])
            via_o = dot(r64["dout"], r["o"].double() - r64["o"])
            print(f"   {tag:15s} d gamma {g:+.6e}  rel err {abs(g - g64) / abs(g64):.2e} | via dout {via_dout / abs(g64):+.2e} "
                  f"via o {via_o / abs(g64):+.2e} | max|dout - dout64|/max {rel(r['dout'], r64['dout']):.1e}, "
                  f"max|o - o64|/max {rel(r['o'], r64['o']):.1e}")
        pg = grads.get(name + ".gamma")
        if pg is not None:
            print(f"   HIP parameter gradient {float(pg):+.6e} (rel err {abs(float(pg) - g64) / abs(g64):.2e}); fp32 reference "
                  f"{float(p32[name + '.gamma'].grad):+.6e}")


if __name__ == "__main__":
    main()
