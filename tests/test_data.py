"""The caller side of the hot path: sample_train_data semantics and the train.py schedules."""
import numpy as np
import pytest


def _utts(rng, n, F=24, lo=130, hi=400):
    return [rng.standard_normal((F, int(rng.integers(lo, hi)))) for _ in range(n)]


def test_draw_crops_consumes_the_rng_like_the_reference(built_lib):
    """same seed -> same utterance order and crop offsets as the restated preprocess.sample_train_data."""
    from oracle.sampling_oracle import sample_train_data
    from voice_converter_cyclegan_amd.data import draw_crops
    rng = np.random.default_rng(0)
    A, B = _utts(rng, 7), _utts(rng, 5)
    np.random.seed(123)
    ref_A, ref_B = sample_train_data(A, B, 128)
    np.random.seed(123)
    ia, sa, ib, sb = draw_crops([u.shape[1] for u in A], [u.shape[1] for u in B], 128)
    assert ref_A.shape == (5, 24, 128) and len(ia) == 5
    for i in range(5):
        assert np.array_equal(ref_A[i], A[ia[i]][:, sa[i]:sa[i] + 128])
        assert np.array_equal(ref_B[i], B[ib[i]][:, sb[i]:sb[i] + 128])
    # and the RNG is left in the same state
    np.random.seed(123); sample_train_data(A, B, 128); x = np.random.randint(10 ** 9)
    np.random.seed(123); draw_crops([u.shape[1] for u in A], [u.shape[1] for u in B], 128); y = np.random.randint(10 ** 9)
    assert x == y


def test_schedule_matches_train_py(built_lib):
    from voice_converter_cyclegan_amd.data import Schedule
    s = Schedule()
    assert s.step(0) == (10, 5, 0.0002, 0.0001)
    assert s.step(10000)[1] == 5 and s.step(10001)[1] == 0                    # train.py:98-99
    assert s.step(200000)[2] == 0.0002
    lc, li, g, d = s.step(200001)                                             # train.py:100-102
    assert abs(g - (0.0002 - 0.0002 / 200000)) < 1e-18 and abs(d - (0.0001 - 0.0001 / 200000)) < 1e-18
    for _ in range(250000):
        lc, li, g, d = s.step(300000)
    assert g == 0 and d == 0                                                  # max(0, ...)


@pytest.mark.gpu
def test_gpu_sampler_equals_oracle_and_drives_training(built_lib):
    import module
    from model import CycleGAN
    from oracle.sampling_oracle import sample_train_data
    from voice_converter_cyclegan_amd.data import TrainDataSampler, Schedule, train_epochs
    rng = np.random.default_rng(1)
    A, B = _utts(rng, 6), _utts(rng, 4)
    smp = TrainDataSampler(A, B, 128)
    np.random.seed(7)
    ref_A, ref_B = sample_train_data(A, B, 128)
    np.random.seed(7)
    got_A, got_B = smp.sample()
    assert got_A.is_cuda and tuple(got_A.shape) == (4, 24, 128)
    assert np.array_equal(got_A.cpu().numpy(), ref_A.astype(np.float32))
    assert np.array_equal(got_B.cpu().numpy(), ref_B.astype(np.float32))
    module.reset_default_graph()
    model = CycleGAN(num_features=24, seed=3, log_dir='/tmp/vcgan_test_log')
    seen = []
    n = train_epochs(model, smp, num_epochs=2, mini_batch_size=2, schedule=Schedule(identity_iterations=2),
                     on_iteration=lambda it, g, d: seen.append((it, float(g), float(d))))
    assert n == 4 and [s[0] for s in seen] == [0, 1, 2, 3] and all(np.isfinite(s[1]) and np.isfinite(s[2]) for s in seen)
    assert model.train_step == 4
    module.reset_default_graph()
