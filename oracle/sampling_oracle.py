"""CPU restatement of preprocess.sample_train_data (/root/reference/preprocess.py:207-238) -- TEST INFRASTRUCTURE ONLY.
numpy is the reference's own dependency here, so unlike the TF path this one restates the exact call sequence on the
global numpy RNG: identical seeds draw identical crops."""
import numpy as np


def sample_train_data(dataset_A, dataset_B, n_frames=128):
    num_samples = min(len(dataset_A), len(dataset_B))                 # :209
    train_data_A_idx = np.arange(len(dataset_A))                      # :210
    train_data_B_idx = np.arange(len(dataset_B))
    np.random.shuffle(train_data_A_idx)                               # :212
    np.random.shuffle(train_data_B_idx)
    A_sub, B_sub = train_data_A_idx[:num_samples], train_data_B_idx[:num_samples]
    out_A, out_B = [], []
    for ia, ib in zip(A_sub, B_sub):                                  # :220
        a = dataset_A[ia]
        assert a.shape[1] >= n_frames
        sa = np.random.randint(a.shape[1] - n_frames + 1)             # :224
        out_A.append(a[:, sa:sa + n_frames])
        b = dataset_B[ib]
        assert b.shape[1] >= n_frames
        sb = np.random.randint(b.shape[1] - n_frames + 1)             # :231
        out_B.append(b[:, sb:sb + n_frames])
    return np.array(out_A), np.array(out_B)                           # :235-238
