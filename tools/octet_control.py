"""The control experiments of profiles/r05_octet_control.txt (bf16 data path, dY in octet forms).  Run on the GPU box:
    python tools/octet_control.py            # (2) back-to-back steps, caller's stream synchronised only, generation outputs requested: the open finding
    python tools/octet_control.py devsync    # (3) device-wide synchronisation after every step, no generation outputs: everything agrees to 3e-6
    rocprofv3 --kernel-trace ... -- python tools/octet_control.py     # next: which reader of generator_B2A's chain starts before its producer ends
Sequence: production(5) control(5) control(0) production(0) production(0) control(0), lambda_identity in brackets; rel-L2 of every gradient tensor
against the first control(0) step, worst three."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge; ge.build()
from tests.test_step_gpu import _setup, _is_noise_bias
from voice_converter_cyclegan_amd._lib import lib, check
devsync = 'devsync' in sys.argv[1:]
B = 8
O, P, A, Bm, eng = _setup(B, gap=False)
eng.set_math('bf16')
check(lib.vc_tune_conv_bf16_override(128, 4, 1)); check(lib.vc_tune_igemm_override(64, 1, 1)); check(lib.vc_tune_wgrad_override(64, 1, 1))
gA = torch.empty((B, 24, 128), device='cuda:0')


def run(mode, lam):
    check(lib.vc_tune_bf16_pixel_octets(mode))
    if devsync:
        eng.step_grads(A, Bm, 10.0, lam); torch.cuda.synchronize()
    else:
        eng.step_grads(A, Bm, 10.0, lam, gen_A=gA)
    return eng.tf_gradients()


def cmp(a, b, tag):
    w = sorted(((float(np.linalg.norm(a[n].astype(np.float64) - b[n].astype(np.float64)) / (np.linalg.norm(b[n].astype(np.float64)) + 1e-30)), n)
                for n in b if not _is_noise_bias(n)), reverse=True)[:3]
    print('%-34s' % tag, [(n, '%.2e' % e) for e, n in w], flush=True)


p5 = run(1, 5.0); c5 = run(0, 5.0)
cmp(p5, c5, 'production(5) vs control(5):')
c0 = run(0, 0.0); p0 = run(1, 0.0); p0b = run(1, 0.0); c0b = run(0, 0.0)
cmp(p0, c0, 'production(0) vs first control(0):'); cmp(p0b, c0, 'production(0) again vs it:'); cmp(c0b, c0, 'control(0) again vs it:'); cmp(p0b, p0, 'production(0) again vs production(0):')
