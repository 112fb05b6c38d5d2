"""Helpers shared by the -m gpu parity tests: call the C-ABI through ctypes on
torch-owned device memory and convert to/from the oracle's channels-last layout."""
import numpy as np
import torch

import vcgan_pkg  # noqa: F401
from voice_converter_cyclegan_amd import _lib
from voice_converter_cyclegan_amd._lib import lib, ptr, check

DEV = 'cuda:0'


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).to(DEV).contiguous()


def stream():
    return _lib.vp(torch.cuda.current_stream().cuda_stream)


def rel_err(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30))


def conv_fwd(x, w, b, kh, kw, sh, sw):
    """x [G,B,Cin,H,W], w [G,kh,kw,Cin,Cout], b [G,Cout] (device tensors) -> y"""
    G, B, Cin, H, W = x.shape
    Cout = w.shape[-1]
    OH, OW = -(-H // sh), -(-W // sw)
    y = torch.full((G, B, Cout, OH, OW), float('nan'), device=DEV)
    check(lib.vc_conv2d_fwd(ptr(x), x[0].numel(), ptr(w), w[0].numel(), ptr(b), b[0].numel() if b is not None else 0,
                            ptr(y), y[0].numel(), G, B, Cin, H, W, Cout, kh, kw, sh, sw, stream()), 'conv_fwd')
    return y


def conv_dgrad(dy, w, H, W, kh, kw, sh, sw, addend=None, init=None):
    G, B, Cout, OH, OW = dy.shape
    Cin = w.shape[-2]
    acc = init is not None
    dx = init.clone() if acc else torch.full((G, B, Cin, H, W), float('nan'), device=DEV)
    check(lib.vc_conv2d_dgrad(ptr(dy), dy[0].numel(), ptr(w), w[0].numel(), ptr(dx), dx[0].numel(),
                              ptr(addend), addend[0].numel() if addend is not None else 0, int(acc),
                              G, B, Cin, H, W, Cout, kh, kw, sh, sw, stream()), 'conv_dgrad')
    return dx


def conv_wgrad(x, dy, kh, kw, sh, sw, init=None):
    G, B, Cin, H, W = x.shape
    Cout = dy.shape[2]
    dw = init.clone() if init is not None else torch.zeros((G, kh, kw, Cin, Cout), device=DEV)
    check(lib.vc_conv2d_wgrad(ptr(x), x[0].numel(), ptr(dy), dy[0].numel(), ptr(dw), dw[0].numel(),
                              G, B, Cin, H, W, Cout, kh, kw, sh, sw, stream()), 'conv_wgrad')
    return dw
