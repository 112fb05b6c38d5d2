"""CPU: every convolution launch of the configurations bench.py reports takes its tile configuration from an EXACT row of the
measured tables (csrc/vc_tuned_igemm.h, csrc/vc_tuned_wgrad.h) -- the headline (fp32, batch 8, 24x128: BASELINE configs[1]),
configs[3]'s per-GPU share (batch 32) and configs[4]'s (36x256, batch 64).  Checked through the host-only shape queries
vc_tune_query_igemm / vc_tune_query_wgrad (no launch, no GPU): a layer-shape change, a table edit or a change of the table
keys that silently drops a benched launch to the nearest-neighbour rule or the cost model fails here.
tests/test_tuned_tables_gpu.py runs the same launches against the oracle (reference model.py:110-125, module.py:148-213)."""
import ctypes as C

import pytest

from tests.step_shapes import step_launches, launch_keys, wgrad_key

BENCHED = [(24, 128, 8), (24, 128, 32), (36, 256, 64), (24, 128, 1), (24, 128, 2), (24, 128, 4), (24, 128, 16), (24, 128, 64)]       # bench.py: headline and its extra.configs (batch 32, 36x256 batch 64, batch 1 = the reference's own); 2, 4, 16, 64: the sizes between
SRC = {0: 'cost model', 1: 'exact row', 2: 'nearest measured size', 3: 'forced', 4: 'bf16 rule'}


@pytest.mark.parametrize('F,T,B', BENCHED)
def test_benched_shapes_hit_their_measured_rows(built_lib, F, T, B):
    lib = built_lib.lib
    out = (C.c_int * 4)()
    missing = []
    for name, op, g, batches in step_launches(F, T, B):
        if op == 'wgrad':
            kt, N, nch = wgrad_key(*g, batches)
            assert lib.vc_tune_query_wgrad(kt, N, nch, 2, 1, out) == 0
            if out[3] != 1:
                missing.append((name, op, (kt, N, nch), SRC[out[3]]))
        else:
            for at, M, P, kr in launch_keys(op, *g, batches[0]):
                assert lib.vc_tune_query_igemm(at, M, P, kr, 2, out) == 0
                if out[3] != 1:
                    missing.append((name, op, (at, M, P, kr), SRC[out[3]]))
    assert not missing, missing


def test_unmeasured_sizes_fall_back_in_order(built_lib):
    """a batch between two measured ones borrows the same layer's row (source 2); a layer never measured goes to the cost model"""
    lib = built_lib.lib
    out = (C.c_int * 4)()
    (at, M, P, kr), = launch_keys('fwd', 512, 1, 32, 2048, 1, 3, 1, 1, 20)       # residual h1 at a batch nobody measured
    assert lib.vc_tune_query_igemm(at, M, P, kr, 2, out) == 0 and out[3] == 2
    assert lib.vc_tune_query_igemm(0, 96, 4096, 320, 1, out) == 0 and out[3] == 0
    assert out[0] in (64, 128) and out[1] in (1, 2, 4) and out[2] >= 1
    assert lib.vc_tune_query_igemm(0, 96, 4096, 321, 1, out) != 0                   # K rows must be whole chunks


def test_table_keys_are_unique():
    """igemm_config / wgrad_config take the FIRST row whose key matches: a second row with the same key is dead, and which
    measurement wins would depend on section order (172 duplicated keys, 61 with conflicting configurations)."""
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'voice-converter-cyclegan_amd', 'csrc')
    for fn in ('vc_tuned_igemm.h', 'vc_tuned_wgrad.h'):
        keys = []
        for l in open(os.path.join(root, fn)):
            m = re.match(r'^\s*\{([-\d,\s]+)\},\s*$', l)
            if m:
                keys.append(tuple(int(x) for x in m.group(1).split(','))[:5])
        assert len(keys) > 50, fn
        dup = {k for k in keys if keys.count(k) > 1}
        assert not dup, (fn, sorted(dup)[:5])
