"""CycleGANEngine: owns the flat fp32 parameter / gradient / Adam buffers on one
GPU and drives the C-ABI hot path (include/vc_cyclegan.h).

It is the object behind `model.CycleGAN` (reference: model.py:9-150): one engine ==
one reference `tf.Session` with its four variable scopes `generator_A2B`,
`generator_B2A`, `discriminator_A`, `discriminator_B` (model.py:44-54).

torch is used for device memory, streams and (to bootstrap the native RCCL communicator, and as the
exchange on non-RCCL backends) torch.distributed; every arithmetic op of the step runs in
libvcgan_hip.so.  There is no CPU fallback.
"""
import collections
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import lib, ptr, check
from .layout import ParamLayout, GENERATOR, DISCRIMINATOR

ADAM_BETA1, ADAM_BETA2, ADAM_EPS = 0.5, 0.999, 1e-8      # model.py:107-108 + TF defaults
SCOPES = ('generator_A2B', 'generator_B2A', 'discriminator_A', 'discriminator_B')
LOSS_NAMES = ('cycle_loss', 'identity_loss', 'generator_loss_A2B', 'generator_loss_B2A', 'generator_loss',
              'discriminator_loss_A', 'discriminator_loss_B', 'discriminator_loss')   # model.py:155-167
MATH_MODES = {'fp32': 0, 'bf16': 1, 'bf16x3': 2}
EXCHANGE_MODES = {'bucketed': 0, 'single': 1}       # vc_plan_set_exchange_mode


def set_default_math(mode):
    """Arithmetic of the operator-level calls of THIS thread (vc_conv2d_*, which carry no plan) and the initial mode of
    plans it creates afterwards (vc_set_math_mode).  Engines set the mode of their own plans: CycleGANEngine.set_math."""
    check(lib.vc_set_math_mode(MATH_MODES[mode]), 'vc_set_math_mode')


class GradSync:
    """Data-parallel gradient exchange: one SUM all-reduce per flat segment, the 1/world factor folded into the Adam
    kernel (grad_scale).  On the 'nccl' backend the collective itself is issued by libvcgan_hip.so on its own RCCL
    communicator (vc_comm_init; torch.distributed only carries the 128-byte unique id to the other ranks); on any other
    backend (gloo in the CPU / shared-GPU tests) torch.distributed does the all-reduce."""

    def __init__(self, group=None, enabled=True, device=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        on = enabled and dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if on else 1
        self.rank = dist.get_rank(group) if on else 0
        self.comm = None                       # native RCCL communicator (vc_comm*)
        if self.world > 1 and device is not None and dist.get_backend(group) == 'nccl':
            ident = (C.c_ubyte * 128)()
            rc0 = lib.vc_comm_unique_id(ident) if self.rank == 0 else 0
            # rank 0's outcome travels WITH the id: a failure there must not leave the other ranks waiting in the broadcast
            box = [(rc0, bytes(ident), lib.vc_last_error().decode() if rc0 else '')]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            if box[0][0] != 0:
                raise _lib.VcError('vc_comm_unique_id failed on rank 0 (rc=%d): %s' % (box[0][0], box[0][2]))
            ident = (C.c_ubyte * 128).from_buffer_copy(box[0][1])
            torch.cuda.set_device(device)
            h = _lib.vp()
            rc = lib.vc_comm_init(self.rank, self.world, ident, C.byref(h))
            # every rank must take the same path: agree on the outcome (MIN over ranks) before using the communicator
            ok = torch.tensor([1 if rc == 0 and lib.vc_comm_world(h) == self.world else 0], device=device, dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
            if int(ok.item()) == 1:
                self.comm = h
            else:
                import warnings
                warnings.warn('native RCCL communicator unavailable on at least one rank (%s); gradient exchange falls back to '
                              'torch.distributed' % (lib.vc_last_error().decode() if rc else 'rank count mismatch'))
                if rc == 0:
                    lib.vc_comm_destroy(h)

    def close(self):
        if self.comm is not None:
            lib.vc_comm_destroy(self.comm)
            self.comm = None

    @property
    def scale(self):
        return 1.0 / self.world

    def all_reduce(self, t, async_op=False, stream=None):
        if self.world == 1:
            return None
        if self.comm is not None and t.is_cuda and t.dtype == torch.float32:
            s = stream if stream is not None else _lib.vp(torch.cuda.current_stream(t.device).cuda_stream)
            check(lib.vc_allreduce_f32(self.comm, ptr(t), t.numel(), s), 'vc_allreduce_f32')
            return None
        return self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def broadcast(self, t, src=0):
        if self.world > 1:
            self.dist.broadcast(t, src=src, group=self.group)

    def barrier(self):
        if self.world > 1:
            self.dist.barrier(group=self.group)


class PlanCache:
    """LRU of native plans keyed by n_frames.  CycleGAN.test sees a new utterance length on almost every call
    (train.py:132, convert.py:46); evicted plans are destroyed (streams, events, geometry tables)."""

    def __init__(self, create, capacity):
        self.create, self.capacity, self.plans = create, capacity, collections.OrderedDict()

    def get(self, T):
        p = self.plans.get(T)
        if p is not None:
            self.plans.move_to_end(T)
            return p
        while len(self.plans) >= self.capacity:
            torch.cuda.synchronize()           # nothing in flight may still use the plan's streams / tables
            _, old = self.plans.popitem(last=False)
            lib.vc_plan_destroy(old)
        p = self.plans[T] = self.create(T)
        return p

    def values(self):
        return self.plans.values()

    def clear(self):
        for p in self.plans.values():
            lib.vc_plan_destroy(p)
        self.plans.clear()


class CycleGANEngine:
    MATH_MODES = MATH_MODES

    def __init__(self, num_features=24, device='cuda:0', seed=None, group=None, data_parallel=True):
        if not torch.cuda.is_available():
            raise RuntimeError('CycleGANEngine needs an MI355X: the hot path is HIP-only (no CPU fallback)')
        self.F = int(num_features)
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.lg = ParamLayout(GENERATOR, self.F)
        self.ld = ParamLayout(DISCRIMINATOR, self.F)
        nG, nD = self.lg.n_floats, self.ld.n_floats
        self.offsets = (0, nG, 2 * nG, 2 * nG + nD)
        self.n_flat = 2 * nG + 2 * nD
        self.nG2 = 2 * nG
        self.params = torch.zeros(self.n_flat, device=self.device)
        self.grads = torch.zeros(self.n_flat, device=self.device)
        self.m = torch.zeros(self.n_flat, device=self.device)
        self.v = torch.zeros(self.n_flat, device=self.device)
        self.losses = torch.zeros(8, device=self.device)
        self._math = 'fp32'
        self._det = False
        self._overlap = True
        self._exchange = 'bucketed'
        # single GPU: replay the whole step as one hipGraph (vc_train_step_graph).  Off by default: measured on ROCm 7.2 the
        # replay is SLOWER than launching the step's ~330 kernels on its five streams (B=1 3.49 vs 3.21 ms, B=8 8.06 vs 7.67)
        self.use_graph = False
        self._gbuf = {}                # (B, T) -> persistent input / output staging of the captured step
        self._plans = PlanCache(lambda T: self._new_plan(T, False), 4)
        self._infer_plans = PlanCache(lambda T: self._new_plan(T, True), 8)
        self._ws = None
        self.sync = GradSync(group, enabled=data_parallel, device=self.device)
        self._comm_stream = torch.cuda.Stream(device=self.device)      # gradient exchange + Adam, beside the backward
        self.exchange_standin = None      # bench.py --dp-selftest: callable(grad_slice) issued on the update stream in place of the collective
        self.measure_exposed = False      # bench.py: record events around the exposed part of exchange + update
        self._exposed = []
        # tf.train.AdamOptimizer keeps beta1_power/beta2_power as fp32 variables (one pair per optimizer)
        self.b1p = [np.float32(ADAM_BETA1), np.float32(ADAM_BETA1)]
        self.b2p = [np.float32(ADAM_BETA2), np.float32(ADAM_BETA2)]
        self.init_parameters(np.random.default_rng(seed))
        self.sync.broadcast(self.params, 0)       # identical replicas

    def close(self):
        """release the native plans (streams, events, geometry tables, filter images) and the RCCL communicator.  The packed-
        operand workspaces of the bf16 mode belong to the process (vc_release_workspaces frees them explicitly)."""
        torch.cuda.synchronize()
        self._plans.clear()
        self._infer_plans.clear()
        self.sync.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ parameters
    def init_parameters(self, rng):
        blocks = [self.lg.init_block(rng), self.lg.init_block(rng), self.ld.init_block(rng), self.ld.init_block(rng)]
        self.params.copy_(torch.from_numpy(np.concatenate(blocks)))

    def _layout(self, i):
        return self.lg if i < 2 else self.ld

    def load_tf_variables(self, tf_vars):
        """{'generator_A2B/h1_conv/kernel': ndarray, ...} (TF names/shapes) -> device."""
        blocks = [self._layout(i).pack(tf_vars, s) for i, s in enumerate(SCOPES)]
        self.params.copy_(torch.from_numpy(np.concatenate(blocks)))

    def _unpack(self, flat_t):
        flat = flat_t.detach().cpu().numpy()
        out = {}
        for i, s in enumerate(SCOPES):
            lay = self._layout(i)
            out.update(lay.unpack(flat[self.offsets[i]:self.offsets[i] + lay.n_floats], s))
        return out

    def tf_variables(self):
        return self._unpack(self.params)

    def tf_gradients(self):
        return self._unpack(self.grads)

    # ------------------------------------------------------------------ plans / workspace
    def _new_plan(self, T, inference):
        h = _lib.vp()
        create = lib.vc_plan_create_inference if inference else lib.vc_plan_create
        check(create(self.F, int(T), C.byref(h)), 'vc_plan_create')
        assert lib.vc_flat_param_floats(h) == self.n_flat
        check(lib.vc_plan_set_math_mode(h, MATH_MODES[self._math]), 'vc_plan_set_math_mode')
        check(lib.vc_plan_set_deterministic(h, int(self._det)), 'vc_plan_set_deterministic')
        if not self._overlap:
            lib.vc_plan_set_overlap(h, 0)
        check(lib.vc_plan_set_exchange_mode(h, EXCHANGE_MODES[self._exchange]), 'vc_plan_set_exchange_mode')
        return h

    def plan(self, T):
        return self._plans.get(int(T))

    def set_overlap(self, enabled):
        """run the step's independent chains on two streams (default) or serialised (for per-kernel timing)"""
        self._overlap = bool(enabled)
        for h in self._plans.values():
            lib.vc_plan_set_overlap(h, int(self._overlap))

    def set_exchange_mode(self, mode):
        """'bucketed' (default): the gradient exchange as 5 segments under the backward; 'single': ONE all-reduce of the flat gradient
        after it (BASELINE.json's literal form) -- vc_plan_set_exchange_mode; same arithmetic, the A/B partner on multi-GPU hardware"""
        if mode not in EXCHANGE_MODES:
            raise ValueError('exchange mode must be one of %s' % sorted(EXCHANGE_MODES))
        self._exchange = mode
        for h in self._plans.values():
            check(lib.vc_plan_set_exchange_mode(h, EXCHANGE_MODES[mode]), 'vc_plan_set_exchange_mode')

    def set_math(self, mode):
        """Arithmetic of the conv kernels of THIS engine's plans (vc_plan_set_math_mode): 'fp32' (default, the reference's
        precision), 'bf16' (bf16 MFMA operands, fp32 accumulate / storage / statistics; BASELINE config 3) or 'bf16x3'."""
        if mode not in MATH_MODES:
            raise ValueError('math mode must be one of %s' % sorted(MATH_MODES))
        self._math = mode
        for cache in (self._plans, self._infer_plans):
            for h in cache.values():
                check(lib.vc_plan_set_math_mode(h, MATH_MODES[mode]), 'vc_plan_set_math_mode')

    def get_math(self):
        return self._math

    def set_deterministic(self, on):
        """vc_plan_set_deterministic: every reduction of this engine's launches in a fixed order (no split-K atomics, ordered
        bias / gamma / beta sums): two runs on identical inputs -- and a hipGraph replay vs launch by launch -- are bitwise
        equal.  Off by default (split-K fills the chip at small batches)."""
        self._det = bool(on)
        for cache in (self._plans, self._infer_plans):
            for h in cache.values():
                check(lib.vc_plan_set_deterministic(h, int(self._det)), 'vc_plan_set_deterministic')

    def _workspace(self, n):
        if self._ws is None or self._ws.numel() < n:
            self._ws = torch.empty(int(n), device=self.device)
        return self._ws

    def _stream(self):
        return _lib.vp(torch.cuda.current_stream(self.device).cuda_stream)

    def _to_dev(self, a):
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=torch.float32).contiguous()
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)

    # ------------------------------------------------------------------ the hot path
    def step_grads(self, A, Bm, lambda_cycle, lambda_identity, gen_A=None, gen_B=None):
        """forward + every gradient of one train step (model.py:44-95) on the current stream."""
        A, Bm = self._to_dev(A), self._to_dev(Bm)
        assert A.shape == Bm.shape and A.dim() == 3 and A.shape[1] == self.F, (A.shape, Bm.shape)
        B, _, T = A.shape
        p = self.plan(T)
        n = lib.vc_step_workspace_floats(p, B)
        ws = self._workspace(n)
        check(lib.vc_cyclegan_step_grads(p, ptr(self.params), ptr(self.grads), ptr(A), ptr(Bm), B,
                                         float(lambda_cycle), float(lambda_identity), ptr(self.losses),
                                         ptr(gen_A), ptr(gen_B), ptr(ws), ws.numel(), self._stream()),
              'vc_cyclegan_step_grads')
        self._keep = (A, Bm)     # keep inputs alive until the stream has consumed them
        return p

    _prof_on = False

    @classmethod
    def profiling(cls, on):
        """per-launch HIP-event timing of the MFMA kernels (vc_prof_enable); captured steps are bypassed while it is on"""
        cls._prof_on = bool(on)
        lib.vc_prof_enable(int(bool(on)))

    def _profiling(self):
        return CycleGANEngine._prof_on

    def _alpha(self, lr, k):
        one = np.float32(1)
        return float(np.float32(lr) * np.sqrt(one - self.b2p[k]) / (one - self.b1p[k]))

    def _advance_beta_powers(self):
        for k in range(2):
            self.b1p[k] = np.float32(self.b1p[k] * np.float32(ADAM_BETA1))
            self.b2p[k] = np.float32(self.b2p[k] * np.float32(ADAM_BETA2))

    def apply_adam(self, generator_learning_rate, discriminator_learning_rate):
        """ApplyAdam over both optimizers' variable lists (model.py:107-108)."""
        s, gs = self._stream(), float(self.sync.scale)
        for k, (lo, hi, lr) in enumerate(((0, self.nG2, generator_learning_rate),
                                          (self.nG2, self.n_flat, discriminator_learning_rate))):
            check(lib.vc_adam_tf_update(ptr(self.params[lo:hi]), ptr(self.m[lo:hi]), ptr(self.v[lo:hi]),
                                        ptr(self.grads[lo:hi]), hi - lo, self._alpha(lr, k),
                                        ADAM_BETA1, ADAM_BETA2, ADAM_EPS, gs, s), 'vc_adam_tf_update')
        self._advance_beta_powers()

    def grad_segments(self, plan):
        """[(kind, index, [(lo, hi), ...])] in the order the segments become final inside step_grads:
        the discriminator half first, then the generator buckets (each bucket = one range per generator)."""
        if self._exchange == 'single':
            return [('all', 0, [(0, self.n_flat)])]
        segs = [('d', 0, [(self.nG2, self.n_flat)])]
        lo, hi = _lib.i64(0), _lib.i64(0)
        for k in range(lib.vc_step_num_g_buckets(plan)):
            check(lib.vc_step_g_bucket(plan, k, C.byref(lo), C.byref(hi)), 'vc_step_g_bucket')
            segs.append(('g', k, [(self.offsets[0] + lo.value, self.offsets[0] + hi.value),
                                  (self.offsets[1] + lo.value, self.offsets[1] + hi.value)]))
        return segs

    def exchange_grads(self, plan):
        """The one gradient exchange of the step (SUM all-reduce of the flat fp32 gradient) on the caller's stream, as
        segments in the order they become final.  Used with set_overlap(False) (per-kernel timing, debugging)."""
        if self.sync.world == 1:
            return
        for kind, k, ranges in self.grad_segments(plan):
            for lo, hi in ranges:
                self.sync.all_reduce(self.grads[lo:hi])

    def exchange_and_update(self, plan, generator_learning_rate, discriminator_learning_rate):
        """Gradient exchange (data parallel) + ApplyAdam, segment by segment on the update stream, each as soon as the
        native schedule says the segment's gradients are final AND its weights are no longer read: the discriminator half
        rides under the whole generator backward, generator buckets under the rest of it.  Adam is HBM-bound
        (28 B/param), the backward MFMA-bound, so they share the chip well; only the last small bucket (h1, d1, d2) is
        exposed.  Arithmetic is identical to exchange_grads() + apply_adam().
        RCCL backend (or a single GPU): ONE native call, vc_step_exchange_and_update, which issues ncclAllReduce per
        segment on the update stream.  Any other backend: the same loop here, with torch.distributed as the collective."""
        main = torch.cuda.current_stream(self.device)
        side = self._comm_stream
        alphas = (self._alpha(generator_learning_rate, 0), self._alpha(discriminator_learning_rate, 1))
        cs = _lib.vp(side.cuda_stream)
        if self.measure_exposed:
            self._ev_bwd = torch.cuda.Event(enable_timing=True); self._ev_bwd.record(main)     # the step's own work on the main stream ends here
        if self.exchange_standin is None and (self.sync.world == 1 or self.sync.comm is not None):
            check(lib.vc_step_exchange_and_update(plan, self.sync.comm, ptr(self.params), ptr(self.grads), ptr(self.m),
                                                  ptr(self.v), alphas[0], alphas[1], ADAM_BETA1, ADAM_BETA2, ADAM_EPS, cs),
                  'vc_step_exchange_and_update')
        else:
            gs = float(self.sync.scale)
            with torch.cuda.stream(side):
                for kind, k, ranges in self.grad_segments(plan):
                    if kind in ('d', 'all'):
                        check(lib.vc_step_wait_d_grads(plan, cs), 'vc_step_wait_d_grads')
                    if kind == 'all':
                        for kk in range(lib.vc_step_num_g_buckets(plan)):
                            check(lib.vc_step_wait_g_bucket(plan, kk, cs), 'vc_step_wait_g_bucket')
                    elif kind == 'g':
                        check(lib.vc_step_wait_g_bucket(plan, k, cs), 'vc_step_wait_g_bucket')
                    if self.exchange_standin is not None:      # bench.py --dp-selftest: stand-in traffic where the collective would run
                        for lo, hi in ranges:
                            self.exchange_standin(self.grads[lo:hi])
                        ws = []
                    else:
                        ws = [self.sync.all_reduce(self.grads[lo:hi], async_op=True) for lo, hi in ranges]
                    for w in ws:
                        w.wait()                     # makes the side stream wait for the collective
                    if kind == 'all':              # one exchange, then the two optimizers' ranges
                        ranges, kinds = [(0, self.nG2), (self.nG2, self.n_flat)], ('g', 'd')
                    else:
                        kinds = (kind,) * len(ranges)
                    for (lo, hi), kd in zip(ranges, kinds):
                        check(lib.vc_adam_tf_update(ptr(self.params[lo:hi]), ptr(self.m[lo:hi]), ptr(self.v[lo:hi]),
                                                    ptr(self.grads[lo:hi]), hi - lo, alphas[1 if kd == 'd' else 0],
                                                    ADAM_BETA1, ADAM_BETA2, ADAM_EPS, gs, cs), 'vc_adam_tf_update')
        main.wait_stream(side)
        if self.measure_exposed:
            ev = torch.cuda.Event(enable_timing=True); ev.record(main)
            self._exposed.append((self._ev_bwd, ev))
        self._advance_beta_powers()

    def exposed_update_ms(self):
        """bench.py: mean time per step between the end of the step's own backward work on the main stream and the end of the
        exchange + Adam on the update stream, i.e. the part of the gradient exchange / update NOT hidden under the backward
        (measure_exposed = True records one event pair per step; synchronises)."""
        torch.cuda.synchronize()
        ts = [a.elapsed_time(b) for a, b in self._exposed]
        self._exposed = []
        return float(np.mean(ts)) if ts else None

    def _graph_step(self, A, Bm, lambda_cycle, lambda_identity, lr_g, lr_d, gen_A, gen_B):
        """the whole step as ONE hipGraph replay (vc_train_step_graph): inputs are staged into persistent buffers (a
        captured graph holds addresses), the scalars that change per step travel through a device buffer"""
        A, Bm = self._to_dev(A), self._to_dev(Bm)
        assert A.shape == Bm.shape and A.dim() == 3 and A.shape[1] == self.F, (A.shape, Bm.shape)
        B, _, T = A.shape
        p = self.plan(T)
        buf = self._gbuf.get((B, T))
        if buf is None:
            mk = lambda: torch.empty((B, self.F, T), device=self.device)
            buf = self._gbuf[(B, T)] = (mk(), mk(), mk(), mk())
        sA, sB, oA, oB = buf
        sA.copy_(A); sB.copy_(Bm)
        ws = self._workspace(lib.vc_step_workspace_floats(p, B))
        check(lib.vc_train_step_graph(p, ptr(self.params), ptr(self.grads), ptr(self.m), ptr(self.v), ptr(sA), ptr(sB), B,
                                      float(lambda_cycle), float(lambda_identity), self._alpha(lr_g, 0), self._alpha(lr_d, 1),
                                      ADAM_BETA1, ADAM_BETA2, ADAM_EPS, ptr(self.losses), ptr(oA), ptr(oB),
                                      ptr(ws), ws.numel(), self._stream()), 'vc_train_step_graph')
        if gen_A is not None:
            gen_A.copy_(oA)
        if gen_B is not None:
            gen_B.copy_(oB)
        self._advance_beta_powers()

    def train_step(self, A, Bm, lambda_cycle, lambda_identity, generator_learning_rate,
                   discriminator_learning_rate, sync=True, gen_A=None, gen_B=None):
        """model.CycleGAN.train (model.py:110-125): returns (generator_loss, discriminator_loss).
        sync=False returns the 8-element device tensor of losses without a host round trip."""
        if self.use_graph and self._overlap and self.sync.world == 1 and not self._profiling():
            self._graph_step(A, Bm, lambda_cycle, lambda_identity, generator_learning_rate, discriminator_learning_rate, gen_A, gen_B)
        else:
            p = self.step_grads(A, Bm, lambda_cycle, lambda_identity, gen_A, gen_B)
            if self._overlap:
                self.exchange_and_update(p, generator_learning_rate, discriminator_learning_rate)
            else:                                        # everything on the caller's stream (per-kernel timing, debugging)
                self.exchange_grads(p)
                self.apply_adam(generator_learning_rate, discriminator_learning_rate)
        if not sync:
            return self.losses
        l = self.losses.cpu().numpy()
        if self.sync.world > 1:        # scalars are per-shard means -> global mean
            t = self.losses.clone(); self.sync.all_reduce(t); l = (t / self.sync.world).cpu().numpy()
        self.last_losses = dict(zip(LOSS_NAMES, (float(x) for x in l)))
        return float(l[4]), float(l[7])

    def generate(self, inputs, direction):
        """model.CycleGAN.test (model.py:128-137): [B, F, T] -> np.ndarray [B, F, T]."""
        if direction == 'A2B':
            off = self.offsets[0]
        elif direction == 'B2A':
            off = self.offsets[1]
        else:
            raise Exception('Conversion direction must be specified.')
        x = self._to_dev(inputs)
        assert x.dim() == 3 and x.shape[1] == self.F, x.shape
        B, _, T = x.shape
        p = self._infer_plans.get(int(T))
        save = self._workspace(lib.vc_generator_save_floats(p, B))
        y = torch.empty_like(x)
        check(lib.vc_generator_forward(p, ptr(self.params[off:]), 0, ptr(x), 0, ptr(y), 0, ptr(save), 1, B,
                                       self._stream()), 'vc_generator_forward')
        return y.cpu().numpy()

    # ------------------------------------------------------------------ checkpoint state
    def state_dict(self):
        return dict(params=self.params.cpu().numpy(), m=self.m.cpu().numpy(), v=self.v.cpu().numpy(),
                    b1p=np.array(self.b1p, np.float32), b2p=np.array(self.b2p, np.float32),
                    num_features=self.F)

    def load_state_dict(self, sd):
        assert int(sd['num_features']) == self.F
        self.params.copy_(torch.from_numpy(np.asarray(sd['params'], np.float32)))
        self.m.copy_(torch.from_numpy(np.asarray(sd['m'], np.float32)))
        self.v.copy_(torch.from_numpy(np.asarray(sd['v'], np.float32)))
        self.b1p = [np.float32(x) for x in sd['b1p']]
        self.b2p = [np.float32(x) for x in sd['b2p']]
