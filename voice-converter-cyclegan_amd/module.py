"""Host mirror of the reference's network definitions (reference module.py).

`generator_gatedcnn(inputs, reuse=False, scope_name=...)` and
`discriminator(inputs, reuse=False, scope_name=...)` keep the reference's names,
argument meaning and error behaviour (module.py:148,188): `inputs` is
[batch, num_features, time]; `scope_name` names a weight set in the default
variable store; `reuse=False` creates it and fails if it already exists (TF:
"Variable ... already exists"), `reuse=True` looks it up and fails if missing.

Instead of adding TF ops to a graph they run the HIP networks eagerly through the
C-ABI (vc_generator_forward / vc_discriminator_forward) and return a torch tensor
on the GPU that carries autograd: backward calls vc_*_backward.  torch is only the
tensor container and the autograd tape.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import lib, ptr, check
from .layout import ParamLayout, GENERATOR, DISCRIMINATOR


_pending_destroy = []          # native plans whose last Python reference is gone; destroyed at the next safe point


def _drain_pending_plans():
    """Destroy the plans queued by _PlanHandle.__del__.  Called where no stream capture can be in progress on this thread (the
    top of VariableStore.plan, interpreter exit): vc_plan_destroy synchronises the device and frees memory, both illegal
    inside a capture (vc_train_step_graph runs one in thread-local capture mode)."""
    if not _pending_destroy:
        return
    torch.cuda.synchronize()                       # nothing in flight may still use the plans' streams / tables
    while _pending_destroy:
        lib.vc_plan_destroy(C.c_void_p(_pending_destroy.pop()))


def _drain_at_exit():
    # interpreter teardown: the HIP runtime may already be shutting down -- a failed synchronise must not turn into a traceback
    try:
        if torch.cuda.is_available():
            _drain_pending_plans()
    except Exception:
        pass


import atexit  # noqa: E402
atexit.register(_drain_at_exit)


class _PlanHandle(C.c_void_p):
    """A native plan that outlives its last user only until the next safe point: the store's LRU holds one reference, every
    autograd node that still needs the plan for its backward holds another (ctx.plan).  __del__ can run wherever the garbage
    collector pleases -- also while another call is capturing a step graph -- so it only QUEUES the handle."""

    def __del__(self):
        try:
            if self.value:
                _pending_destroy.append(self.value)
        except Exception:
            pass


class VariableStore:
    """The analogue of TF's default graph variable collection: scope name -> flat fp32 block."""
    PLAN_CAPACITY = 8            # geometries kept alive; CycleGAN.test sees a new utterance length on almost every call

    def __init__(self):
        import collections
        self.scopes = {}          # name -> dict(kind, layout, params (leaf tensor), num_features)
        self._plans = collections.OrderedDict()

    def create(self, name, kind, num_features, device, rng=None, storage=None):
        if name in self.scopes:
            raise ValueError('Variable %s/h1_conv/kernel already exists, disallowed. Did you mean to set '
                             'reuse=True in VarScope?' % name)
        lay = ParamLayout(kind, num_features)
        if storage is None:
            storage = torch.zeros(lay.n_floats, device=device)
            storage.copy_(torch.from_numpy(lay.init_block(rng or np.random.default_rng())))
        assert storage.numel() == lay.n_floats and storage.is_contiguous()
        storage.requires_grad_(True)
        self.scopes[name] = dict(kind=kind, layout=lay, params=storage, num_features=num_features)
        return self.scopes[name]

    def get(self, name):
        if name not in self.scopes:
            raise ValueError('Variable %s/h1_conv/kernel does not exist, or was not created with '
                             'tf.get_variable(). Did you mean to set reuse=None in VarScope?' % name)
        return self.scopes[name]

    def plan(self, F, T):
        _drain_pending_plans()
        key = (F, T)
        h = self._plans.get(key)
        if h is not None:
            self._plans.move_to_end(key)
            return h
        while len(self._plans) >= self.PLAN_CAPACITY:
            self._plans.popitem(last=False)        # queued for destruction once no pending backward refers to it (_PlanHandle.__del__)
        _drain_pending_plans()                     # ... and destroyed NOW if nothing does (this is a safe point): never PLAN_CAPACITY + 1 plans alive
        h = _PlanHandle()
        check(lib.vc_plan_create(F, int(T), C.byref(h)), 'vc_plan_create')
        self._plans[key] = h
        return h

    def trainable_variables(self, substring):
        """model.py:92-95 partitions variables by 'generator' / 'discriminator' in the name."""
        return {n: s for n, s in self.scopes.items() if substring in n}


_default_store = VariableStore()


def get_variable_store():
    return _default_store


def reset_default_graph():
    """tf.reset_default_graph(): forget every variable scope."""
    global _default_store
    _default_store = VariableStore()
    return _default_store


def _stream(t):
    return _lib.vp(torch.cuda.current_stream(t.device).cuda_stream)


class _GeneratorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, params, plan):
        B, F, T = x.shape
        x = x.contiguous()
        save = torch.empty(lib.vc_generator_save_floats(plan, B), device=x.device)
        y = torch.empty_like(x)
        check(lib.vc_generator_forward(plan, ptr(params), 0, ptr(x), 0, ptr(y), 0, ptr(save), 1, B, _stream(x)),
              'vc_generator_forward')
        ctx.plan = plan
        ctx.save_for_backward(x, params, save)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, params, save = ctx.saved_tensors
        B = x.shape[0]
        dy = dy.contiguous()
        need_dx, need_dp = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dx = torch.empty_like(x) if need_dx else None
        dp = torch.zeros_like(params) if need_dp else None
        scr = torch.empty(lib.vc_scratch_floats(ctx.plan, B), device=x.device)
        check(lib.vc_generator_backward(ctx.plan, ptr(params), 0, ptr(x), 0, ptr(dy), 0, ptr(save), ptr(dp), 0,
                                        ptr(dx), 0, 0, ptr(scr), scr.numel(), 1, B, _stream(x)), 'vc_generator_backward')
        return dx, dp, None


class _DiscriminatorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, params, plan):
        B = x.shape[0]
        x = x.contiguous()
        hw = lib.vc_discriminator_out_elems(plan)
        save = torch.empty(lib.vc_discriminator_save_floats(plan, B), device=x.device)
        o = torch.empty((B, hw), device=x.device)
        check(lib.vc_discriminator_forward(plan, ptr(params), 0, ptr(x), 0, ptr(o), 0, ptr(save), 1, B, _stream(x)),
              'vc_discriminator_forward')
        ctx.plan = plan
        ctx.save_for_backward(x, params, save)
        return o

    @staticmethod
    def backward(ctx, do):
        x, params, save = ctx.saved_tensors
        B = x.shape[0]
        do = do.contiguous()
        need_dx, need_dp = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dx = torch.empty_like(x) if need_dx else None
        dp = torch.zeros_like(params) if need_dp else None
        scr = torch.empty(lib.vc_scratch_floats(ctx.plan, B), device=x.device)
        check(lib.vc_discriminator_backward(ctx.plan, ptr(params), 0, ptr(x), 0, ptr(do), 0, ptr(save), ptr(dp), 0,
                                            ptr(dx), 0, 0, ptr(scr), scr.numel(), 1, B, _stream(x)), 'vc_discriminator_backward')
        return dx, dp, None


def _as_input(inputs, device=None):
    if isinstance(inputs, torch.Tensor):
        t = inputs
        if not t.is_cuda:
            t = t.to(device or 'cuda:0')
        return t.to(torch.float32)
    return torch.from_numpy(np.ascontiguousarray(inputs, dtype=np.float32)).to(device or 'cuda:0')


def _apply(kind, fn, inputs, reuse, scope_name):
    store = get_variable_store()
    x = _as_input(inputs, store.scopes[scope_name]['params'].device if scope_name in store.scopes else None)
    if x.dim() != 3:
        raise ValueError('inputs must be [batch_size, num_features, time], got %s' % (tuple(x.shape),))
    F, T = int(x.shape[1]), int(x.shape[2])
    if reuse:
        sc = store.get(scope_name)
    else:
        sc = store.create(scope_name, kind, F, x.device)
    if sc['kind'] != kind or sc['num_features'] != F:
        raise ValueError('scope %s holds a different network / num_features' % scope_name)
    return fn.apply(x, sc['params'], store.plan(F, T)), x


def generator_gatedcnn(inputs, reuse=False, scope_name='generator_gatedcnn'):
    """reference module.py:148-185.  [B, F, T] -> [B, F, T] (T % 4 == 0)."""
    y, _ = _apply(GENERATOR, _GeneratorFn, inputs, reuse, scope_name)
    return y


def discriminator(inputs, reuse=False, scope_name='discriminator'):
    """reference module.py:188-213.  [B, F, T] -> [B, H', W', 1] sigmoid patch map."""
    o, x = _apply(DISCRIMINATOR, _DiscriminatorFn, inputs, reuse, scope_name)
    B, F, T = x.shape
    hp = -(-F // 4)                    # H: stride 1, 2, 2, 1 (SAME padding)
    return o.view(B, hp, o.shape[1] // hp, 1)
