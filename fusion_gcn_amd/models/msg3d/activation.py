"""Activation by name (reference torch_src/models/msg3d/activation.py:10-20).  The modules are markers inside the reference-shaped
module tree (they keep the Sequential indices of the state-dict keys); the HIP ops apply the activation themselves, so only
``relu`` and ``linear`` -- the two the MS-G3D model uses -- are backed by kernels."""
import torch.nn as nn

_FACTORIES = {"relu": lambda inplace: nn.ReLU(inplace=inplace), "leakyrelu": lambda inplace: nn.LeakyReLU(0.2, inplace=inplace),
              "tanh": lambda inplace: nn.Tanh(), "linear": lambda inplace: nn.Identity(), None: lambda inplace: nn.Identity()}


def activation_factory(name, inplace=True):
    if name not in _FACTORIES:
        raise ValueError("Not supported activation:", name)
    return _FACTORIES[name](inplace)


def is_relu(module) -> bool:
    """What the fused ops need to know about an activation marker; anything but ReLU / Identity has no kernel."""
    if isinstance(module, nn.ReLU):
        return True
    if isinstance(module, nn.Identity):
        return False
    raise NotImplementedError(f"{type(module).__name__}: the HIP MS-G3D ops implement relu and linear activations")
