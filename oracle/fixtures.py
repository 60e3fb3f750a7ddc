"""Deterministic inputs and weights shared by the golden-vector generator, the
tests, `__graft_entry__.smoke()` and `bench.py`.

TEST INFRASTRUCTURE.  Nothing in the product path (`iccv2025-gdl_amd/`) imports
this module.  Everything here is numpy-only (PCG64 streams are identical on
every machine), so the GPU box regenerates bit-identical inputs without the
reference being present.

The weight fill follows the *shape* of the reference's initialisation
(`utils/utils.py:15-23`: conv kaiming-normal fan_out, linear xavier-normal) but
keys every tensor's random stream on its state_dict name, and deliberately uses
non-trivial BatchNorm gamma/beta and running statistics so that a wrong
scale/shift/momentum path cannot hide behind gamma=1, beta=0.
"""
import zlib

import numpy as np

N_CLASSES = {"VGGSound": 309, "KineticSound": 34, "kinect400": 400, "CREMAD": 6, "AVE": 28}
# /root/reference/models/basic_model.py:15-26


def _rng(name, salt=0):
    return np.random.default_rng([zlib.crc32(name.encode()), salt])


def named_tensor(name, shape):
    """Deterministic float32 tensor for state_dict entry `name`."""
    shape = tuple(int(s) for s in shape)
    r = _rng(name)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return np.zeros(shape, dtype=np.int64)
    if leaf == "running_mean":
        return (0.05 * r.standard_normal(shape)).astype(np.float32)
    if leaf == "running_var":
        return (1.0 + 0.1 * np.abs(r.standard_normal(shape))).astype(np.float32)
    if len(shape) == 4:  # conv weight [K, C, R, S]; kaiming_normal_(fan_out, relu)
        k, c, rr, ss = shape
        std = np.sqrt(2.0 / (k * rr * ss))
        return (std * r.standard_normal(shape)).astype(np.float32)
    if len(shape) == 2:  # linear weight [out, in]; xavier_normal_
        std = np.sqrt(2.0 / (shape[0] + shape[1]))
        return (std * r.standard_normal(shape)).astype(np.float32)
    # 1-D: BN / LayerNorm gamma / beta, or a linear bias
    is_bn = ("bn" in name) or ("downsample.1" in name) or ("norm" in name)
    if leaf == "weight" and is_bn:
        return (1.0 + 0.1 * r.standard_normal(shape)).astype(np.float32)
    if leaf == "bias" and is_bn:
        return (0.1 * r.standard_normal(shape)).astype(np.float32)
    return (0.01 * r.standard_normal(shape)).astype(np.float32)


def make_state(shapes):
    """`shapes`: ordered {name: shape} -> ordered {name: ndarray}."""
    return {k: named_tensor(k, v) for k, v in shapes.items()}


def make_batch(seed, batch, spec_hw, frames, image_hw, n_classes):
    """Synthetic batch with the layout the reference datasets emit
    (`dataset/CramedDataset.py:57-110`): spec [B,F,T'] f32, image [B,3,T,H,W] f32,
    label [B] int64.  N(0,1) values as BASELINE.md section 4 prescribes."""
    r = np.random.default_rng([1234, seed])
    spec = r.standard_normal((batch,) + tuple(spec_hw), dtype=np.float32)
    image = r.standard_normal((batch, 3, frames) + tuple(image_hw), dtype=np.float32)
    label = r.integers(0, n_classes, size=(batch,), dtype=np.int64)
    return spec, image, label


# ---------------------------------------------------------------- ResNet18 topology
def resnet18_param_shapes(prefix, in_ch):
    """Ordered parameter shapes of the reference encoder
    (`models/backbone.py:75-156`), in `named_parameters()` order."""
    out = {}
    out[prefix + "conv1.weight"] = (64, in_ch, 7, 7)
    out[prefix + "bn1.weight"] = (64,)
    out[prefix + "bn1.bias"] = (64,)
    inpl = 64
    for li, planes in enumerate((64, 128, 256, 512), start=1):
        for bi in range(2):
            p = f"{prefix}layer{li}.{bi}."
            stride = 2 if (bi == 0 and li > 1) else 1
            out[p + "conv1.weight"] = (planes, inpl, 3, 3)
            out[p + "bn1.weight"] = (planes,)
            out[p + "bn1.bias"] = (planes,)
            out[p + "conv2.weight"] = (planes, planes, 3, 3)
            out[p + "bn2.weight"] = (planes,)
            out[p + "bn2.bias"] = (planes,)
            if stride != 1 or inpl != planes:
                out[p + "downsample.0.weight"] = (planes, inpl, 1, 1)
                out[p + "downsample.1.weight"] = (planes,)
                out[p + "downsample.1.bias"] = (planes,)
            inpl = planes
    return out


def resnet18_buffer_shapes(prefix):
    out = {}

    def bn(p, c):
        out[p + "running_mean"] = (c,)
        out[p + "running_var"] = (c,)
        out[p + "num_batches_tracked"] = ()

    bn(prefix + "bn1.", 64)
    inpl = 64
    for li, planes in enumerate((64, 128, 256, 512), start=1):
        for bi in range(2):
            p = f"{prefix}layer{li}.{bi}."
            stride = 2 if (bi == 0 and li > 1) else 1
            bn(p + "bn1.", planes)
            bn(p + "bn2.", planes)
            if stride != 1 or inpl != planes:
                bn(p + "downsample.1.", planes)
            inpl = planes
    return out


def model_param_shapes(n_classes, fusion="concat_dgl"):
    """`AVClassifier_DGL.named_parameters()` order: fusion head first, then
    audio_net, then visual_net (`models/basic_model.py:29-44`)."""
    out = {}
    if fusion == "concat_dgl":  # fusion_modules.py:45-49
        out["fusion_module.fc_out.weight"] = (n_classes, 1024)
        out["fusion_module.fc_out.bias"] = (n_classes,)
        out["fusion_module.fc_auxi.weight"] = (n_classes, 1024)
        out["fusion_module.fc_auxi.bias"] = (n_classes,)
    elif fusion == "sum_dgl":  # fusion_modules.py:16-20
        out["fusion_module.fc_x.weight"] = (n_classes, 512)
        out["fusion_module.fc_x.bias"] = (n_classes,)
        out["fusion_module.fc_y.weight"] = (n_classes, 512)
        out["fusion_module.fc_y.bias"] = (n_classes,)
    elif fusion == "gated_dgl":  # fusion_modules.py:219-224
        out["fusion_module.fc_x.weight"] = (512, 512)
        out["fusion_module.fc_x.bias"] = (512,)
        out["fusion_module.fc_y.weight"] = (512, 512)
        out["fusion_module.fc_y.bias"] = (512,)
        out["fusion_module.fc_out.weight"] = (n_classes, 512)
        out["fusion_module.fc_out.bias"] = (n_classes,)
    elif fusion == "film_dgl":  # fusion_modules.py:132-138
        out["fusion_module.fc.weight"] = (512, 512 * 512)
        out["fusion_module.fc.bias"] = (512,)
        out["fusion_module.fc_out.weight"] = (n_classes, 512)
        out["fusion_module.fc_out.bias"] = (n_classes,)
    elif fusion == "concat":  # fusion_modules.py:33-36
        out["fusion_module.fc_out.weight"] = (n_classes, 1024)
        out["fusion_module.fc_out.bias"] = (n_classes,)
    else:
        raise NotImplementedError(fusion)
    out.update(resnet18_param_shapes("audio_net.", 1))
    out.update(resnet18_param_shapes("visual_net.", 3))
    return out


def model_buffer_shapes():
    out = {}
    out.update(resnet18_buffer_shapes("audio_net."))
    out.update(resnet18_buffer_shapes("visual_net."))
    return out


def model_state(n_classes, fusion="concat_dgl"):
    ps = make_state(model_param_shapes(n_classes, fusion))
    bs = make_state(model_buffer_shapes())
    return ps, bs


# ---------------------------------------------------------------- Swin topology ("next" row N4)
# /root/reference/models/swin_transformer.py:486-560: constructor arguments of the two configurations the tests use.
# The reference's defaults are Swin-B; Swin-T's settings are passed explicitly.
SWIN_T = dict(img=224, patch=4, embed=96, depths=(2, 2, 6, 2), heads=(3, 6, 12, 24), window=7, mlp=4)
SWIN_TINY2 = dict(img=56, patch=4, embed=96, depths=(2, 2), heads=(3, 6), window=7, mlp=4)  # 14x14 -> 7x7 tokens


def swin_param_shapes(cfg, prefix=""):
    """Ordered {name: shape} of SwinTransformer.named_parameters() (args.pe = 0, ape = False, patch_norm = True):
    swin_transformer.py:463-478 (patch embed), :98-122 (attention), :205-215 (block), :321-323 (merging), :556."""
    E, p = cfg["embed"], cfg["patch"]
    res = cfg["img"] // p
    sh = {prefix + "patch_embed.proj.weight": (E, 3, p, p), prefix + "patch_embed.proj.bias": (E,),
          prefix + "patch_embed.norm.weight": (E,), prefix + "patch_embed.norm.bias": (E,)}
    nl = len(cfg["depths"])
    for i, (depth, nh) in enumerate(zip(cfg["depths"], cfg["heads"])):
        dim, r = E << i, res >> i
        ws = min(cfg["window"], r)
        for j in range(depth):
            b = f"{prefix}layers.{i}.blocks.{j}."
            sh[b + "norm1.weight"] = (dim,)
            sh[b + "norm1.bias"] = (dim,)
            sh[b + "attn.relative_position_bias_table"] = ((2 * ws - 1) ** 2, nh)
            sh[b + "attn.qkv.weight"] = (3 * dim, dim)
            sh[b + "attn.qkv.bias"] = (3 * dim,)
            sh[b + "attn.proj.weight"] = (dim, dim)
            sh[b + "attn.proj.bias"] = (dim,)
            sh[b + "norm2.weight"] = (dim,)
            sh[b + "norm2.bias"] = (dim,)
            sh[b + "mlp.fc1.weight"] = (cfg["mlp"] * dim, dim)
            sh[b + "mlp.fc1.bias"] = (cfg["mlp"] * dim,)
            sh[b + "mlp.fc2.weight"] = (dim, cfg["mlp"] * dim)
            sh[b + "mlp.fc2.bias"] = (dim,)
        if i < nl - 1:
            d = f"{prefix}layers.{i}.downsample."
            sh[d + "reduction.weight"] = (2 * dim, 4 * dim)
            sh[d + "norm.weight"] = (4 * dim,)
            sh[d + "norm.bias"] = (4 * dim,)
    sh[prefix + "norm.weight"] = (E << (nl - 1),)
    sh[prefix + "norm.bias"] = (E << (nl - 1),)
    return sh


def swin_input(cfg, batch, frames, seed=0):
    """[B, 3, T, img, img] float32 frames (the 'visual' modality layout, swin_transformer.py:598-601)."""
    return np.random.default_rng([4321, seed]).standard_normal((batch, 3, frames, cfg["img"], cfg["img"]), dtype=np.float32)


def swin_dgl_state(n_classes, cfg):
    """Parameters and buffers of the Swin composition (ResNet18 audio + Swin visual + ConcatFusion_DGL over 512 + C):
    ({name: array}, {name: array}) in named_parameters() / buffer order."""
    feat = cfg["embed"] << (len(cfg["depths"]) - 1)
    sh = {"fusion_module.fc_out.weight": (n_classes, 512 + feat), "fusion_module.fc_out.bias": (n_classes,),
          "fusion_module.fc_auxi.weight": (n_classes, 512 + feat), "fusion_module.fc_auxi.bias": (n_classes,)}
    sh.update(resnet18_param_shapes("audio_net.", 1))
    sh.update(swin_param_shapes(cfg, "visual_net."))
    return make_state(sh), make_state(resnet18_buffer_shapes("audio_net."))
