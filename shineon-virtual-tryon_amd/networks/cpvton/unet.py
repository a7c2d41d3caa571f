"""U-Net generator of the try-on module (reference: models/networks/cpvton/unet.py:9-211).

Same constructor signatures and the same nn.Sequential layout as the reference (so checkpoints drop
in); down-steps are Conv4x4/s2, up-steps are activation -> bilinear x2 -> Conv3x3 -> norm, attention is
spent from the innermost block outwards with one SelfAttention on the down side and one on the up side
of every attended block.
"""
from torch import nn

from ... import ops
from ..attention.sagan import SelfAttention
from ..layers import (HipActivation, HipConv2d, HipGELU, HipInstanceNorm2d, HipBatchNorm2d, HipLeakyReLU, HipReLU,
                      HipUpsample2x, Sine, Swish)

_NORMS = {"instance": HipInstanceNorm2d, "batch": HipBatchNorm2d}


def _resolve_norm(norm_layer):
    """Accept our own classes, the names 'instance'/'batch', or the torch classes the reference passes."""
    if norm_layer in (HipInstanceNorm2d, HipBatchNorm2d):
        return norm_layer
    if isinstance(norm_layer, str):
        return _NORMS[norm_layer]
    if norm_layer is nn.InstanceNorm2d:
        return HipInstanceNorm2d
    if norm_layer is nn.BatchNorm2d:
        return HipBatchNorm2d
    raise ValueError(f"unsupported norm_layer {norm_layer}")


class UnetGenerator(nn.Module):
    def __init__(self, input_nc, output_nc, num_downs, num_attention, ngf=64, norm_layer=HipBatchNorm2d,
                 use_dropout=False, use_self_attn=False, activation=None):
        super().__init__()
        if use_dropout:
            raise NotImplementedError("use_dropout is never enabled on the reference's hot path")

        def attn():
            return use_self_attn if use_self_attn and num_attention > 0 else None

        block = UnetSkipConnectionBlock(ngf * 8, ngf * 8, submodule=None, norm_layer=norm_layer, innermost=True,
                                        self_attn=attn(), activation=activation)
        num_attention -= 1
        for _ in range(num_downs - 5):
            block = UnetSkipConnectionBlock(ngf * 8, ngf * 8, submodule=block, norm_layer=norm_layer,
                                            self_attn=attn(), activation=activation)
            num_attention -= 1
        for outer, inner in ((ngf * 4, ngf * 8), (ngf * 2, ngf * 4), (ngf, ngf * 2)):
            block = UnetSkipConnectionBlock(outer, inner, submodule=block, norm_layer=norm_layer,
                                            self_attn=attn(), activation=activation)
            num_attention -= 1
        block = UnetSkipConnectionBlock(output_nc, ngf, input_nc=input_nc, submodule=block, outermost=True,
                                        norm_layer=norm_layer, self_attn=attn(), activation=activation)
        self.model = block

    def forward(self, input):
        return self.model(input)


class UnetSkipConnectionBlock(nn.Module):
    """X ---------------- identity ---------------- X
         |-- downsampling -- |submodule| -- upsampling --|"""

    def __init__(self, outer_nc, inner_nc, input_nc=None, submodule=None, outermost=False, innermost=False,
                 norm_layer=HipBatchNorm2d, self_attn=False, use_dropout=False, activation=None):
        super().__init__()
        norm_layer = _resolve_norm(norm_layer)
        self.outermost = outermost
        use_bias = norm_layer is HipInstanceNorm2d
        if input_nc is None:
            input_nc = outer_nc
        downconv = HipConv2d(input_nc, inner_nc, kernel_size=4, stride=2, padding=1, bias=use_bias)
        down_activation = HipLeakyReLU(0.2) if activation is None else _get_activation_fn(activation)
        downnorm = norm_layer(inner_nc)
        up_activation = HipReLU() if activation is None else _get_activation_fn(activation)
        upnorm = norm_layer(outer_nc)
        # The reference's default down activation is LeakyReLU(0.2, inplace=True): it overwrites the block
        # input before the skip concatenation, so the skip carries leaky_relu(x).
        self.skip_carries_activation = activation is None and not outermost
        up_in = inner_nc if innermost else inner_nc * 2
        upconv = HipConv2d(up_in, outer_nc, kernel_size=3, stride=1, padding=1, bias=use_bias)
        if outermost:
            down = [downconv]
        elif innermost:
            down = [down_activation, downconv]
        else:
            down = [down_activation, downconv, downnorm]
        up = [up_activation, HipUpsample2x(), upconv, upnorm]
        norm_kind = "always" if norm_layer is HipInstanceNorm2d else "train"
        upconv.zero_bias_grad = norm_kind                      # upconv -> upnorm
        if not outermost and not innermost:
            downconv.zero_bias_grad = norm_kind                # downconv -> downnorm
        if self_attn:
            down.append(SelfAttention(inner_nc, "relu"))
            up.append(SelfAttention(outer_nc, "relu"))
        model = down + ([] if innermost else [submodule]) + up
        self.model = nn.Sequential(*model)

    def forward(self, x):
        out = self.forward_pair(x)
        return out.cat() if isinstance(out, SkipPair) else out

    def forward_pair(self, x, preact=None):
        """forward() with the skip concatenation [x | h] left as a SkipPair: the enclosing block's `act -> Upsample` reads
        both parts directly (ops.upsample2x_bilinear_cat), so torch.cat's copy never happens on the hot path.
        preact: this block's down activation of x, already computed by the producer of x (_run: norm -> block)."""
        if self.outermost:
            return _run(list(self.model), x)
        mods = list(self.model)
        if self.skip_carries_activation:
            x = mods[0](x)
            mods = mods[1:]
        elif preact is not None:
            return SkipPair(x, _run(mods[1:], preact))
        elif self.takes_preactivation() and x.is_cuda:
            # x feeds the skip concatenation AND this block's down activation: one node for the pair, so that the backward pass
            # merges the two gradients in the activation's own kernel (ops.fork_act) instead of an accumulation launch
            xs, a = ops.fork_act(x, mods[0].kind, mods[0].param)
            return SkipPair(xs, _run(mods[1:], a))
        return SkipPair(x, _run(mods, x))

    def takes_preactivation(self):
        """The down activation (first module) is a plain, not-in-place activation of the block input: its value can come from
        the kernel that produces the input."""
        first = self.model[0]
        return (not self.outermost and not self.skip_carries_activation and isinstance(first, HipActivation)
                and first.kind in ops.ACT_CODES)


class SkipPair:
    """cat([x, h], 1) of a skip connection, not yet materialised."""

    __slots__ = ("x", "h")

    def __init__(self, x, h):
        self.x, self.h = x, h

    def cat(self):
        return ops.cat_channels([self.x, self.h])


def _run(mods, h):
    """nn.Sequential semantics with one fusion: an activation module directly followed by the bilinear upsample (the up
    path's `act -> Upsample`) runs as a single kernel, which also consumes a pending skip concatenation in place."""
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, HipActivation) and i + 1 < len(mods) and isinstance(mods[i + 1], HipUpsample2x):
            if isinstance(h, SkipPair):
                h = ops.upsample2x_bilinear_cat(h.x, h.h, m.kind, m.param)
            else:
                h = ops.upsample2x_bilinear(h, m.kind, m.param)
            i += 2
        elif (isinstance(m, HipInstanceNorm2d) and i + 1 < len(mods) and isinstance(mods[i + 1], UnetSkipConnectionBlock)
              and mods[i + 1].takes_preactivation() and not isinstance(h, SkipPair) and h.is_cuda):
            # norm -> [submodule: act -> conv ...]: the norm kernel writes act(y) beside y (y stays for the skip connection);
            # (y, act(y)) come back as the two outputs of ONE autograd node (ops.fork_act)
            nxt = mods[i + 1]
            y, a = ops.instance_norm_act(h, m.eps, nxt.model[0].kind, nxt.model[0].param)
            h = nxt.forward_pair(y, preact=a)
            i += 2
        else:
            if isinstance(h, SkipPair):
                h = h.cat()
            h = m.forward_pair(h) if isinstance(m, UnetSkipConnectionBlock) else m(h)
            i += 1
    return h.cat() if isinstance(h, SkipPair) else h


def _get_activation_fn(activation):
    if activation == "relu":
        return HipReLU()
    if activation == "gelu":
        return HipGELU()
    if activation == "swish":
        return Swish()
    if activation == "sine":
        return Sine()
    raise RuntimeError(f"The selected activation should be relu/gelu/swish/sine, not {activation}")
