"""Autograd glue shared by every HIP op: the backward kernels are not themselves differentiable, so a SECOND
differentiation through an op must fail loudly instead of treating the first-order gradient as a constant (which is what
torch does, silently, for a custom Function whose backward is opaque).  The reference never needs it -- its Grad-CAM++
(src/services/XAI/models/cam.py:38-43) calls autograd.grad(..., create_graph=True) and then only squares / cubes the
first-order gradients -- so first-order results under create_graph=True stay exact and free."""
import functools

import torch

from ._lib import MMSkinError


class _NoSecondOrder(torch.autograd.Function):
    """Identity on `grad`, tied to the op's differentiable inputs; differentiating through it raises."""

    @staticmethod
    def forward(ctx, name, grad, *anchors):
        ctx.op_name = name
        return grad.view_as(grad)

    @staticmethod
    def backward(ctx, g):
        raise MMSkinError(
            f"second-order differentiation through the HIP op {ctx.op_name} is not supported: its backward runs opaque "
            "gfx950 kernels.  First-order gradients, also under autograd.grad(..., create_graph=True), are exact.")


def no_second_order(cls):
    """Class decorator for torch.autograd.Function subclasses whose backward launches HIP kernels."""
    fwd, bwd = cls.forward, cls.backward

    @functools.wraps(fwd)
    def forward(ctx, *args):
        ctx._mmskin_anchors = tuple(a for a in args if isinstance(a, torch.Tensor) and a.requires_grad)
        return fwd(ctx, *args)

    @functools.wraps(bwd)
    def backward(ctx, *grads):
        with torch.no_grad():
            outs = bwd(ctx, *[g.detach() if isinstance(g, torch.Tensor) else g for g in grads])
        anchors = ctx._mmskin_anchors + tuple(g for g in grads if isinstance(g, torch.Tensor) and g.requires_grad)
        if not torch.is_grad_enabled() or not anchors:
            return outs
        single = not isinstance(outs, tuple)
        seq = (outs,) if single else outs
        seq = tuple(_NoSecondOrder.apply(cls.__name__, o, *anchors) if isinstance(o, torch.Tensor) else o for o in seq)
        return seq[0] if single else seq

    cls.forward = staticmethod(forward)
    cls.backward = staticmethod(backward)
    return cls
