"""Caller side of the hot path ("next" row SURVEY 8f-1): the per-epoch minibatch assembly of the reference's
train.py:78-111 with the normalised MCEP utterances resident in HBM.

 * TrainDataSampler.sample()  == preprocess.sample_train_data (preprocess.py:207-238): same use of the numpy global RNG
   (two shuffles, then one randint per A/B pair, in that order), so a seeded run draws the same crops as the reference;
   the crops themselves are gathered on the GPU (vc_crop_batch) from one flat device buffer per domain.
 * Schedule == the hyper-parameter schedule of train.py:15-26,98-102.
"""
import numpy as np
import torch

from . import _lib
from ._lib import lib, ptr, check


def draw_crops(len_A, len_B, n_frames=128):
    """The index logic of preprocess.sample_train_data (preprocess.py:209-233) without touching the data.
    len_X[i] = number of frames of utterance i.  Returns (idx_A, start_A, idx_B, start_B)."""
    num_samples = min(len(len_A), len(len_B))
    idx_A = np.arange(len(len_A)); idx_B = np.arange(len(len_B))
    np.random.shuffle(idx_A)
    np.random.shuffle(idx_B)
    idx_A, idx_B = idx_A[:num_samples], idx_B[:num_samples]
    start_A = np.empty(num_samples, np.int32); start_B = np.empty(num_samples, np.int32)
    for i, (a, b) in enumerate(zip(idx_A, idx_B)):
        assert len_A[a] >= n_frames
        start_A[i] = np.random.randint(len_A[a] - n_frames + 1)
        assert len_B[b] >= n_frames
        start_B[i] = np.random.randint(len_B[b] - n_frames + 1)
    return idx_A, start_A, idx_B, start_B


class _Domain:
    def __init__(self, utterances, device):
        self.F = int(utterances[0].shape[0])
        self.len = np.array([u.shape[1] for u in utterances], np.int32)
        self.off = np.concatenate([[0], np.cumsum(self.len.astype(np.int64) * self.F)[:-1]]).astype(np.int64)
        flat = np.concatenate([np.ascontiguousarray(u, dtype=np.float32).reshape(-1) for u in utterances])
        self.data = torch.from_numpy(flat).to(device)


class TrainDataSampler:
    def __init__(self, dataset_A, dataset_B, n_frames=128, device='cuda:0'):
        """dataset_X: list of [num_features, T_i] arrays (coded_sps_*_norm in train.py:50-52)."""
        self.device, self.n_frames = torch.device(device), n_frames
        self.A, self.B = _Domain(dataset_A, self.device), _Domain(dataset_B, self.device)
        assert self.A.F == self.B.F

    def _gather(self, dom, idx, start):
        n = len(idx)
        meta = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(self.device)
        off, ln, st = meta(dom.off[idx], np.int64), meta(dom.len[idx], np.int32), meta(start, np.int32)
        out = torch.empty((n, dom.F, self.n_frames), device=self.device)
        check(lib.vc_crop_batch(ptr(dom.data), ptr(off), ptr(ln), ptr(st), ptr(out), n, dom.F, self.n_frames,
                                _lib.vp(torch.cuda.current_stream(self.device).cuda_stream)), 'vc_crop_batch')
        self._keep = (off, ln, st)
        return out

    def sample(self):
        """-> (train_data_A, train_data_B), two [n_samples, F, n_frames] fp32 DEVICE tensors."""
        ia, sa, ib, sb = draw_crops(self.A.len, self.B.len, self.n_frames)
        return self._gather(self.A, ia, sa), self._gather(self.B, ib, sb)


class Schedule:
    """train.py:15-26 + 98-102.  call step(num_iterations) once per minibatch, like the loop body does."""

    def __init__(self, generator_learning_rate=0.0002, discriminator_learning_rate=0.0001, lambda_cycle=10,
                 lambda_identity=5, identity_iterations=10000, decay_start=200000):
        self.generator_learning_rate = generator_learning_rate
        self.discriminator_learning_rate = discriminator_learning_rate
        self.generator_learning_rate_decay = generator_learning_rate / 200000
        self.discriminator_learning_rate_decay = discriminator_learning_rate / 200000
        self.lambda_cycle, self.lambda_identity = lambda_cycle, lambda_identity
        self.identity_iterations, self.decay_start = identity_iterations, decay_start

    def step(self, num_iterations):
        if num_iterations > self.identity_iterations:
            self.lambda_identity = 0
        if num_iterations > self.decay_start:
            self.generator_learning_rate = max(0, self.generator_learning_rate - self.generator_learning_rate_decay)
            self.discriminator_learning_rate = max(0, self.discriminator_learning_rate - self.discriminator_learning_rate_decay)
        return (self.lambda_cycle, self.lambda_identity, self.generator_learning_rate, self.discriminator_learning_rate)


def train_epochs(model, sampler, num_epochs, mini_batch_size=1, schedule=None, on_iteration=None):
    """The epoch / minibatch loop of train.py:78-111 (without its file I/O): returns the number of iterations run."""
    schedule = schedule or Schedule()
    it = 0
    for epoch in range(num_epochs):
        dataset_A, dataset_B = sampler.sample()
        n_samples = dataset_A.shape[0]
        for i in range(n_samples // mini_batch_size):
            num_iterations = n_samples // mini_batch_size * epoch + i
            lc, li, lr_g, lr_d = schedule.step(num_iterations)
            start, end = i * mini_batch_size, (i + 1) * mini_batch_size
            g, d = model.train(input_A=dataset_A[start:end], input_B=dataset_B[start:end], lambda_cycle=lc,
                               lambda_identity=li, generator_learning_rate=lr_g, discriminator_learning_rate=lr_d)
            if on_iteration:
                on_iteration(num_iterations, g, d)
            it += 1
    return it
