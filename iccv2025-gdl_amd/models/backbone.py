"""Drop-in mirror of /root/reference/models/backbone.py for the MI355X build.

Same public names (`conv3x3`, `conv1x1`, `BasicBlock`, `ResNet`, `resnet18`), constructor
arguments, attribute / parameter / buffer names, registration order and initialisation
(backbone.py:75-132), so `state_dict`s, `model.apply(weight_init)` and seeded initial
weights are interchangeable with the reference.  The torch.nn layers are only parameter
containers: `ResNet.forward` runs the whole encoder as one planned sequence of hand-written
gfx950 kernels (csrc/encoder.cpp) behind a single autograd node.
"""
import os

import torch
import torch.nn as nn

from gdl import _lib as L
from gdl.encoder import EncoderEngine


def _weights_only_conv(cin, cout, k, stride, pad):
    """An nn.Conv2d used purely as a named, correctly shaped and initialised weight holder (bias-free, as every
    convolution of the reference backbone); the arithmetic happens in csrc/."""
    return nn.Conv2d(cin, cout, (k, k), stride=(stride, stride), padding=(pad, pad), bias=False)


def conv3x3(in_planes, out_planes, stride=1, groups=1, dilation=1):
    """Same call signature as backbone.py:20-23; the engine implements groups = dilation = 1 only."""
    if (groups, dilation) != (1, 1):
        raise NotImplementedError("gdl: grouped / dilated 3x3 convolutions are not used by the reference and not built")
    return _weights_only_conv(in_planes, out_planes, 3, stride, 1)


def conv1x1(in_planes, out_planes, stride=1):
    """Same call signature as backbone.py:26-28 (the downsample shortcut)."""
    return _weights_only_conv(in_planes, out_planes, 1, stride, 0)


class BasicBlock(nn.Module):
    """backbone.py:31-68.  Executed by the encoder engine as part of ResNet.forward."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64, dilation=1,
                 norm_layer=None):
        super().__init__()
        bn = nn.BatchNorm2d if norm_layer is None else norm_layer
        if (groups, base_width) != (1, 64):
            raise ValueError('BasicBlock only supports groups=1 and base_width=64')  # same error as the reference
        if dilation > 1:
            raise NotImplementedError("Dilation > 1 not supported in BasicBlock")
        # registration order = state_dict / named_parameters order of the reference block:
        # conv1, bn1, (relu), conv2, bn2, downsample
        for name, mod in (("conv1", conv3x3(inplanes, planes, stride)), ("bn1", bn(planes)), ("relu", nn.ReLU(inplace=True)),
                          ("conv2", conv3x3(planes, planes)), ("bn2", bn(planes)), ("downsample", downsample)):
            setattr(self, name, mod)
        self.stride = stride

    def forward(self, x):
        raise NotImplementedError("gdl: BasicBlock runs inside the ResNet encoder engine; call the ResNet module")


class _ResNetFn(torch.autograd.Function):
    """One autograd node for the whole encoder.  `want` selects the output: 'fmap' is what the
    reference's ResNet.forward returns ([N,512,h,w] float32 NCHW), 'feat' additionally folds the
    pooling glue of basic_model.py:73-82 and returns [B,512]."""

    @staticmethod
    def forward(ctx, net, x, want, *params):
        eng = net._engine(x)
        net._bind(eng)
        training = net.training
        feat, fmap = eng.forward(x, training, want_feat=(want == "feat"), want_fmap=(want == "fmap"))
        ctx.eng, ctx.want, ctx.training = eng, want, training
        ctx.serial = eng.serial
        ctx.pshapes = [p.shape for p in params]
        ctx.pdev = params[0].device
        ctx.set_materialize_grads(False)  # an undefined upstream gradient (the fusion head returned None) costs nothing
        return feat if want == "feat" else fmap

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None, None, None) + (None,) * len(ctx.pshapes)
        eng = ctx.eng
        if not ctx.training:
            raise RuntimeError("gdl: backward through an eval-mode encoder forward is not supported")
        if ctx.serial != eng.serial:
            raise RuntimeError("gdl: the encoder ran another training forward since this graph was built; "
                               "its saved activations are gone")
        flat = torch.empty(sum(eng.param_numel), device=ctx.pdev)
        grads, o = [], 0
        for s, n in zip(ctx.pshapes, eng.param_numel):
            grads.append(flat[o:o + n].view(s))
            o += n
        if ctx.want == "feat":
            eng.backward(grads, dfeat=g)
        else:
            eng.backward(grads, dfmap=g)
        return (None, None, None) + tuple(grads)


class ResNet(nn.Module):

    def __init__(self, args, block, layers, modality, num_classes=1000, pool='avgpool', zero_init_residual=False,
                 groups=1, width_per_group=64, replace_stride_with_dilation=None, norm_layer=None):
        super().__init__()
        bn = nn.BatchNorm2d if norm_layer is None else norm_layer
        rswd = [False] * 3 if replace_stride_with_dilation is None else list(replace_stride_with_dilation)
        if len(rswd) != 3:
            raise ValueError("replace_stride_with_dilation should be None "
                             "or a 3-element tuple, got {}".format(replace_stride_with_dilation))
        if bn is not nn.BatchNorm2d or block is not BasicBlock or list(layers) != [2, 2, 2, 2]:
            raise NotImplementedError("gdl: the MI355X engine implements resnet18 (BasicBlock [2,2,2,2], BatchNorm2d)")
        if any(rswd) or (groups, width_per_group) != (1, 64):
            raise NotImplementedError("gdl: dilation / groups are not used by the reference and not implemented")
        stem_in = {'audio': 1, 'visual': 3}  # backbone.py:96-101
        if modality not in stem_in:
            raise NotImplementedError('Incorrect modality, should be audio or visual but got {}'.format(modality))
        self.modality, self.pool, self.args = modality, pool, args
        self._norm_layer, self.groups, self.base_width, self.dilation = bn, groups, width_per_group, 1
        # modules in the reference's registration order: conv1, bn1, relu, maxpool, layer1..4
        self.inplanes = 64
        self.conv1 = _weights_only_conv(stem_in[modality], 64, 7, 2, 3)
        self.bn1 = bn(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        for idx, (width, stride) in enumerate(((64, 1), (128, 2), (256, 2), (512, 2))):
            setattr(self, "layer%d" % (idx + 1), self._make_layer(block, width, layers[idx], stride=stride))
        # constructor-time initialisation (backbone.py:117-127): it consumes the global RNG in module order, which
        # seeded runs of the reference depend on, so it is reproduced call for call even though the DGL script
        # re-initialises everything through utils.weight_init afterwards
        for mod in self.modules():
            if isinstance(mod, nn.Conv2d):
                nn.init.kaiming_normal_(mod.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(mod, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.normal_(mod.weight, mean=1, std=0.02)
                nn.init.zeros_(mod.bias)
        if zero_init_residual:
            for mod in self.modules():
                if isinstance(mod, BasicBlock):
                    nn.init.zeros_(mod.bn2.weight)
        # engine state (not part of state_dict)
        self.gdl_dtype = os.environ.get("GDL_DTYPE", "bf16")
        self._engines = {}

    def _make_layer(self, block, planes, blocks, stride=1, dilate=False):
        """A stage of `blocks` BasicBlocks; the first one changes resolution / width and then carries the
        conv1x1 + BatchNorm shortcut (backbone.py:134-156)."""
        width = planes * block.expansion
        shortcut = None
        if stride != 1 or self.inplanes != width:
            shortcut = nn.Sequential(conv1x1(self.inplanes, width, stride), self._norm_layer(width))
        stage = [block(self.inplanes, planes, stride, shortcut, self.groups, self.base_width, 1, self._norm_layer)]
        self.inplanes = width
        stage += [block(width, planes, groups=self.groups, base_width=self.base_width, dilation=self.dilation,
                        norm_layer=self._norm_layer) for _ in range(blocks - 1)]
        return nn.Sequential(*stage)

    # ------------------------------------------------------------------ engine plumbing
    def _bn_layers(self):
        out = [self.bn1]
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                out += [blk.bn1, blk.bn2]
                if blk.downsample is not None:
                    out.append(blk.downsample[1])
        return out

    def _engine(self, x):
        if self.modality == 'visual':
            if x.dim() != 5:
                raise RuntimeError("gdl: visual input must be [B,3,T,H,W] (backbone.py:162)")
            B, C, T, H, W = x.shape
            if C != 3:
                raise RuntimeError(f"gdl: visual input has {C} channels, expected 3")
        else:
            if x.dim() != 4 or x.shape[1] != 1:
                raise RuntimeError("gdl: audio input must be [B,1,F,T'] (main_dgl.py:100)")
            B, _, H, W = x.shape
            T = 1
        if not x.is_cuda:
            raise RuntimeError("gdl: the encoder runs on an MI355X only; move the model and inputs to cuda "
                               "(there is no CPU path)")
        key = (B, T, H, W, self.gdl_dtype, x.device.index)
        eng = self._engines.get(key)
        if eng is None:
            eng = EncoderEngine(self.modality, self.gdl_dtype, B, T, H, W, x.device)
            self._engines[key] = eng
        return eng

    def _bind(self, eng):
        bns = self._bn_layers()
        eng.set_params([p.data for p in self.parameters()], [b.running_mean for b in bns],
                       [b.running_var for b in bns], [b.num_batches_tracked for b in bns])

    def _run(self, x, want):
        x = x.float().contiguous()
        return _ResNetFn.apply(self, x, want, *self.parameters())

    def forward(self, x):
        """backbone.py:158-201: returns the un-pooled layer4 map [N,512,h,w] (N = B*T for visual)."""
        return self._run(x, "fmap")

    def forward_pooled(self, x):
        """ResNet.forward followed by the pooling glue of basic_model.py:73-82 -> [B,512]."""
        return self._run(x, "feat")


def _resnet(arch, args, block, layers, modality, progress, **kwargs):
    return ResNet(args, block, layers, modality, **kwargs)


def resnet18(modality, args, progress=True, **kwargs):
    return _resnet('resnet18', args, BasicBlock, [2, 2, 2, 2], modality, progress, **kwargs)
