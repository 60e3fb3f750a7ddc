"""Host-side mirror of the two helpers `main_dgl.py:17` imports from the reference's utils/utils.py
(`setup_seed`, :7-12, and `weight_init`, :15-23): same names, same effect on every RNG and on every
parameter, so seeded runs of the reference script start from identical weights.
"""
import random

import numpy as np
import torch
import torch.nn as nn

_SEEDERS = (torch.manual_seed, torch.cuda.manual_seed_all, np.random.seed, random.seed)


def setup_seed(seed):
    """Seed torch (CPU + every GPU), numpy and `random`; ask for deterministic library kernels.  (On this build
    determinism does not depend on that flag: every reduction of csrc/ runs in a fixed order.)"""
    for seeder in _SEEDERS:
        seeder(seed)
    torch.backends.cudnn.deterministic = True


def _init_linear(m):
    nn.init.xavier_normal_(m.weight)
    nn.init.zeros_(m.bias)


def _init_conv(m):
    nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")


def _init_bn(m):
    nn.init.ones_(m.weight)
    nn.init.zeros_(m.bias)


# first match wins, in the order the reference tests the types
_INITIALISERS = ((nn.Linear, _init_linear), (nn.Conv2d, _init_conv), (nn.BatchNorm2d, _init_bn))


def weight_init(m):
    """`model.apply(weight_init)` (main_dgl.py:238): Xavier-normal Linear weights with zero bias, Kaiming-normal
    (fan_out, relu) convolution weights, BatchNorm gamma = 1 / beta = 0.  Other module types are left alone."""
    for kind, init in _INITIALISERS:
        if isinstance(m, kind):
            init(m)
            return
