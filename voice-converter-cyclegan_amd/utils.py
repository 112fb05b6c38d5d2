"""reference utils.py:6-12 on torch GPU tensors (HIP reductions, autograd-aware)."""
import torch

from . import _lib
from ._lib import lib, ptr, check


class _MeanLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, y_hat, squared):
        y = y.expand_as(y_hat).contiguous().float()
        y_hat = y_hat.contiguous()
        loss = torch.zeros((), device=y_hat.device)
        d = torch.empty_like(y_hat)
        f = lib.vc_l2_loss if squared else lib.vc_l1_loss
        check(f(ptr(y), ptr(y_hat), y_hat.numel(), ptr(loss), ptr(d),
                _lib.vp(torch.cuda.current_stream(y_hat.device).cuda_stream)), 'vc_l%d_loss' % (2 if squared else 1))
        ctx.save_for_backward(d)
        return loss

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        gy = g * d
        return (-gy if ctx.needs_input_grad[0] else None), gy, None


def _t(x, like):
    return x if isinstance(x, torch.Tensor) else torch.as_tensor(x, dtype=torch.float32, device=like.device)


def l1_loss(y, y_hat):
    """tf.reduce_mean(tf.abs(y - y_hat))  (utils.py:6-8)"""
    return _MeanLoss.apply(_t(y, y_hat), y_hat, False)


def l2_loss(y, y_hat):
    """tf.reduce_mean(tf.square(y - y_hat))  (utils.py:10-12)"""
    return _MeanLoss.apply(_t(y, y_hat), y_hat, True)
