"""Host mirror of reference model.py: class CycleGAN with the same constructor,
train/test/save/load/summary signatures and return types (model.py:9,110,128,140,148,153),
so the reference's train.py / convert.py can `from model import CycleGAN` unchanged.

TF graph + Session are replaced by a CycleGANEngine (flat fp32 buffers on one MI355X,
HIP kernels through the C-ABI).  With the default `generator`/`discriminator` callables
the step runs as ONE fused native schedule (vc_cyclegan_step_grads); with user-supplied
callables (the reference's injection seam, model.py:9,14-15) the same loss algebra
(model.py:44-90) is composed from the callables on the autograd tape.
"""
import atexit
import os
import time
from datetime import datetime

import numpy as np
import torch

from . import module as _module
from .module import discriminator, generator_gatedcnn
from .utils import l1_loss, l2_loss
from .engine import CycleGANEngine, SCOPES, LOSS_NAMES
from . import tfio
from . import _lib

G_SUMMARIES = ('cycle_loss', 'identity_loss', 'generator_loss_A2B', 'generator_loss_B2A', 'generator_loss')
D_SUMMARIES = ('discriminator_loss_A', 'discriminator_loss_B', 'discriminator_loss')


class ScalarWriter:
    """Stand-in for tf.summary.FileWriter (model.py:26-29,116,121).  Writes both
      * a TensorBoard event file `events.out.tfevents.<time>.<host>` (one Event per add_summary call of the reference:
        the merged generator summaries, then the merged discriminator summaries; tags `<group>/<name>`), and
      * one CSV per scalar in the layout TensorBoard exports (`Wall time,Step,Value`, same as the reference's
        train_log/*.csv, so train_log/plot.py reads them unchanged)
    under log_dir."""

    def __init__(self, log_dir, flush_every=200, flush_secs=120.0):
        self.log_dir, self.flush_every, self.flush_secs, self.rows, self.n = log_dir, flush_every, flush_secs, {}, 0
        self.events, self.pending, self.last_flush = None, [], time.time()
        atexit.register(self.close)            # tf.summary.FileWriter flushes on a timer and at close; so do we

    def add(self, group, tag, step, value):
        now = time.time()
        self.rows.setdefault((group, tag), []).append((now, step, value))
        if self.pending and self.pending[-1][:2] == (group, step):
            self.pending[-1][3]['%s/%s' % (group, tag)] = value
        else:
            self.pending.append((group, step, now, {'%s/%s' % (group, tag): value}))
        self.n += 1
        if self.n >= self.flush_every or now - self.last_flush >= self.flush_secs:
            self.flush()

    def close(self):
        self.flush()
        if self.events is not None:
            self.events.close()
            self.events = None

    def flush(self):
        self.last_flush = time.time()
        if not self.n:
            return
        os.makedirs(self.log_dir, exist_ok=True)
        if self.events is None:
            self.events = tfio.EventFileWriter(self.log_dir)
        for _, step, wall, scalars in self.pending:
            self.events.add_scalars(step, scalars, wall)
        self.events.flush()
        self.pending = []
        for (group, tag), rows in self.rows.items():
            path = os.path.join(self.log_dir, 'run_%s-tag-%s_%s.csv' % (os.path.basename(self.log_dir), group, tag))
            new = not os.path.exists(path)
            with open(path, 'a') as f:
                if new:
                    f.write('Wall time,Step,Value\n')
                for w, s, v in rows:
                    f.write('%r,%d,%r\n' % (w, s, v))
        self.rows, self.n = {}, 0


class CycleGAN(object):

    def __init__(self, num_features, discriminator=discriminator, generator=generator_gatedcnn,
                 mode='train', log_dir='./log', device=None, seed=None, process_group=None):
        self.num_features = num_features
        self.input_shape = [None, num_features, None]      # [batch_size, num_features, num_frames]
        self.discriminator = discriminator
        self.generator = generator
        self.mode = mode
        self._device = device or ('cuda:%d' % int(os.environ.get('LOCAL_RANK', '0')))
        self._seed, self._group = seed, process_group
        # save() also writes the TF V2 checkpoint files the reference's Saver would (set False to keep only the .npz)
        self.save_tf_checkpoint = os.environ.get('VC_SAVE_TF_CHECKPOINT', '1') != '0'

        self.build_model()
        self.optimizer_initializer()

        # Under the data-parallel launch (one process per GPU, the reference's train.py unchanged on every rank) only
        # rank 0 touches the file system: checkpoints, event files and CSVs have ONE writer.
        self._io_rank = self.engine.sync.rank == 0
        if self.mode == 'train':
            self.train_step = 0
            now = datetime.now()
            self.log_dir = os.path.join(log_dir, now.strftime('%Y%m%d-%H%M%S'))
            self.writer = ScalarWriter(self.log_dir) if self._io_rank else None
            self.generator_summaries, self.discriminator_summaries = self.summary()

    # ------------------------------------------------------------------ model.py:32-100
    def build_model(self):
        """Creates the four variable scopes (first use of each scope name in the reference graph,
        model.py:44-54) as views of one flat buffer and records which path train() takes."""
        self.engine = CycleGANEngine(self.num_features, device=self._device, seed=self._seed, group=self._group)
        store = _module.get_variable_store()
        from .layout import GENERATOR, DISCRIMINATOR
        for i, scope in enumerate(SCOPES):
            lay = self.engine._layout(i)
            view = self.engine.params[self.engine.offsets[i]:self.engine.offsets[i] + lay.n_floats]
            store.create(scope, GENERATOR if i < 2 else DISCRIMINATOR, self.num_features, self.engine.device,
                         storage=view)          # raises like TF if a second CycleGAN is built in one process
        self._native = (self.generator is generator_gatedcnn and self.discriminator is _module.discriminator)
        self.generator_vars = sorted(store.trainable_variables('generator'))            # model.py:95
        self.discriminator_vars = sorted(store.trainable_variables('discriminator'))    # model.py:94

    def optimizer_initializer(self):
        """Two tf.train.AdamOptimizer(lr=<fed per step>, beta1=0.5) (model.py:103-108): the slots and the
        fp32 beta powers live in the engine; learning rates arrive with every train() call."""
        self.generator_optimizer = ('AdamTF', 0.5, 0.999, 1e-8, 'generator')
        self.discriminator_optimizer = ('AdamTF', 0.5, 0.999, 1e-8, 'discriminator')

    # ------------------------------------------------------------------ model.py:110-125
    def train(self, input_A, input_B, lambda_cycle, lambda_identity, generator_learning_rate,
              discriminator_learning_rate):
        if not _lib.lib.vc_thread_is_product_run():
            raise RuntimeError('a test / tuning switch of include/vc_cyclegan_testing.h is active in this thread: '
                               'CycleGAN.train runs with production kernel configurations only')
        if self._native:
            generator_loss, discriminator_loss = self.engine.train_step(
                input_A, input_B, lambda_cycle, lambda_identity, generator_learning_rate, discriminator_learning_rate)
            scal = self.engine.last_losses
        else:
            scal = self._train_with_callables(input_A, input_B, lambda_cycle, lambda_identity,
                                              generator_learning_rate, discriminator_learning_rate)
            generator_loss, discriminator_loss = scal['generator_loss'], scal['discriminator_loss']
        if self.mode == 'train':
            if self.writer is not None:
                for tag in G_SUMMARIES:
                    self.writer.add('generator_summaries', tag, self.train_step, scal[tag])
                for tag in D_SUMMARIES:
                    self.writer.add('discriminator_summaries', tag, self.train_step, scal[tag])
            self.train_step += 1
        return np.float32(generator_loss), np.float32(discriminator_loss)

    def _train_with_callables(self, input_A, input_B, lambda_cycle, lambda_identity, lr_g, lr_d):
        """The reference graph (model.py:44-90) composed from the injected callables on the autograd tape."""
        eng, store = self.engine, _module.get_variable_store()
        A, Bm = eng._to_dev(input_A), eng._to_dev(input_B)
        G, D = self.generator, self.discriminator
        gen_B = G(inputs=A, reuse=True, scope_name='generator_A2B')
        cycle_A = G(inputs=gen_B, reuse=True, scope_name='generator_B2A')
        gen_A = G(inputs=Bm, reuse=True, scope_name='generator_B2A')
        cycle_B = G(inputs=gen_A, reuse=True, scope_name='generator_A2B')
        id_A = G(inputs=A, reuse=True, scope_name='generator_B2A')
        id_B = G(inputs=Bm, reuse=True, scope_name='generator_A2B')
        dA_fake = D(inputs=gen_A, reuse=True, scope_name='discriminator_A')
        dB_fake = D(inputs=gen_B, reuse=True, scope_name='discriminator_B')
        cycle_loss = l1_loss(y=A, y_hat=cycle_A) + l1_loss(y=Bm, y_hat=cycle_B)
        identity_loss = l1_loss(y=A, y_hat=id_A) + l1_loss(y=Bm, y_hat=id_B)
        g_A2B = l2_loss(y=torch.ones_like(dB_fake), y_hat=dB_fake)
        g_B2A = l2_loss(y=torch.ones_like(dA_fake), y_hat=dA_fake)
        generator_loss = g_A2B + g_B2A + lambda_cycle * cycle_loss + lambda_identity * identity_loss
        g_params = [store.get(n)['params'] for n in ('generator_A2B', 'generator_B2A')]
        g_grads = torch.autograd.grad(generator_loss, g_params)
        fake_A, fake_B = gen_A.detach(), gen_B.detach()          # model.py:118-119 feeds the fetched arrays
        dA_real = D(inputs=A, reuse=True, scope_name='discriminator_A')
        dB_real = D(inputs=Bm, reuse=True, scope_name='discriminator_B')
        dA_f = D(inputs=fake_A, reuse=True, scope_name='discriminator_A')
        dB_f = D(inputs=fake_B, reuse=True, scope_name='discriminator_B')
        d_loss_A = (l2_loss(y=torch.ones_like(dA_real), y_hat=dA_real) + l2_loss(y=torch.zeros_like(dA_f), y_hat=dA_f)) / 2
        d_loss_B = (l2_loss(y=torch.ones_like(dB_real), y_hat=dB_real) + l2_loss(y=torch.zeros_like(dB_f), y_hat=dB_f)) / 2
        discriminator_loss = d_loss_A + d_loss_B
        d_params = [store.get(n)['params'] for n in ('discriminator_A', 'discriminator_B')]
        d_grads = torch.autograd.grad(discriminator_loss, d_params)
        for i, g in enumerate(list(g_grads) + list(d_grads)):
            eng.grads[eng.offsets[i]:eng.offsets[i] + g.numel()].copy_(g)
        if eng.sync.world > 1:
            eng.sync.all_reduce(eng.grads)
        with torch.no_grad():
            eng.apply_adam(lr_g, lr_d)
        vals = [cycle_loss, identity_loss, g_A2B, g_B2A, generator_loss, d_loss_A, d_loss_B, discriminator_loss]
        return dict(zip(LOSS_NAMES, (float(v.detach()) for v in vals)))

    # ------------------------------------------------------------------ model.py:128-137
    def test(self, inputs, direction):
        if direction == 'A2B':
            generation = self.engine.generate(inputs, 'A2B')
        elif direction == 'B2A':
            generation = self.engine.generate(inputs, 'B2A')
        else:
            raise Exception('Conversion direction must be specified.')
        return generation

    # ------------------------------------------------------------------ model.py:140-150
    def save(self, directory, filename):
        """tf.train.Saver().save over ALL global variables: weights, Adam slots, beta powers (model.py:21).
        One .npz keyed by the TF variable names (`<var>`, `<var>/Adam`, `<var>/Adam_1`, `beta{1,2}_power{,_1}`)."""
        eng = self.engine
        path = os.path.join(directory, filename)
        if not self._io_rank:                  # replicas are identical: rank 0 writes, everybody leaves together
            eng.sync.barrier()
            return path
        try:                                   # whatever happens here, the other ranks are waiting in the barrier
            os.makedirs(directory, exist_ok=True)
            out = {}
            out.update(eng.tf_variables())
            out.update({k + '/Adam': v for k, v in eng._unpack(eng.m).items()})
            out.update({k + '/Adam_1': v for k, v in eng._unpack(eng.v).items()})
            # optimizer creation order in model.py:107-108: discriminator first, then generator
            out['beta1_power'], out['beta2_power'] = eng.b1p[1], eng.b2p[1]
            out['beta1_power_1'], out['beta2_power_1'] = eng.b1p[0], eng.b2p[0]
            tmp = path + '.npz.tmp'
            with open(tmp, 'wb') as f:
                np.savez(f, **out)
            os.replace(tmp, path + '.npz')         # a reader never sees a half-written checkpoint
            if self.save_tf_checkpoint:
                # the files tf.train.Saver writes (minus the .meta graph): `<path>.index`, `<path>.data-00000-of-00001`, `checkpoint`
                tfio.write_tf_checkpoint(path, out)
            if self.mode == 'train' and self.writer is not None:
                self.writer.flush()
        finally:
            eng.sync.barrier()
        return path

    def load(self, filepath):
        """tf.train.Saver().restore: `<filepath>.npz` (ours) or a TF V2 checkpoint `<filepath>.index` + `.data-*` (ours or
        one written by the reference, e.g. its published pre-trained model); a checkpoint without optimizer slots (or a
        model built with mode='test', whose graph has no optimizers) restores the weights only."""
        eng = self.engine
        if os.path.exists(filepath + '.npz'):
            with np.load(filepath + '.npz') as z:
                d = {k: z[k] for k in z.files}
        elif tfio.is_tf_checkpoint(filepath):
            d = tfio.read_tf_checkpoint(filepath)
        else:
            raise IOError('no checkpoint at %s(.npz|.index)' % filepath)
        eng.load_tf_variables(d)
        if self.mode != 'train' or not any(k.endswith('/Adam') for k in d):
            return
        need = ['beta1_power', 'beta2_power', 'beta1_power_1', 'beta2_power_1']
        for i, sc in enumerate(SCOPES):
            for r in eng._layout(i).recs:
                need += [sc + '/' + r['name'] + '/Adam', sc + '/' + r['name'] + '/Adam_1']
        missing = [k for k in need if k not in d]
        if missing:
            import warnings
            warnings.warn('checkpoint %s has optimizer slots but lacks %d of them (first: %s): restoring the weights only'
                          % (filepath, len(missing), missing[0]))
            return
        blocks_m, blocks_v = [], []
        for i, s in enumerate(SCOPES):
            lay = eng._layout(i)
            names = [s + '/' + r['name'] for r in lay.recs]
            blocks_m.append(lay.pack({n: d[n + '/Adam'] for n in names}, s))
            blocks_v.append(lay.pack({n: d[n + '/Adam_1'] for n in names}, s))
        eng.m.copy_(torch.from_numpy(np.concatenate(blocks_m)))
        eng.v.copy_(torch.from_numpy(np.concatenate(blocks_v)))
        eng.b1p = [np.float32(d['beta1_power_1']), np.float32(d['beta1_power'])]
        eng.b2p = [np.float32(d['beta2_power_1']), np.float32(d['beta2_power'])]

    # ------------------------------------------------------------------ model.py:153-169
    def summary(self):
        return (['generator_summaries/' + t for t in G_SUMMARIES],
                ['discriminator_summaries/' + t for t in D_SUMMARIES])


if __name__ == '__main__':
    model = CycleGAN(num_features=24)
    print('Graph Compile Successeded.')
