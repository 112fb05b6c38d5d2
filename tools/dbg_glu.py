# debugging aid: the GLU parity cases of tests/test_ops_gpu.py in pytest's order, several rounds, with an error map on failure
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge; ge.build()
from tests import gpu_util as U
from voice_converter_cyclegan_amd._lib import lib, ptr, check
from oracle import cyclegan_oracle as O
REFS = {}
def run(B, T, mode, packed, seed, rnd):
    rng = np.random.default_rng(seed)
    G, C = 2, 128
    x = rng.standard_normal((G, B, T, 24)).astype(np.float32)
    w = (rng.standard_normal((G, 15, 24, 2 * C)) / np.sqrt(360)).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, (G, 2 * C)).astype(np.float32)
    check(lib.vc_set_math_mode({'fp32': 0, 'bf16': 1}[mode]))
    key = (B, T, mode)
    if key not in REFS:
        O.set_operand_rounding('bf16' if mode == 'bf16' else None)
        ref = []
        for g in range(G):
            v, _ = O.conv1d_fwd(x[g], w[g][:, :, :C], b[g][:C], 1)
            gt, _ = O.conv1d_fwd(x[g], w[g][:, :, C:], b[g][C:], 1)
            ref.append(np.transpose(O.glu_fwd(v, gt)[0], (0, 2, 1)))
        REFS[key] = np.stack(ref)
        O.set_operand_rounding(None)
    ref = REFS[key]
    x_d = U.dev(np.ascontiguousarray(np.transpose(x, (0, 1, 3, 2)))); w_d, b_d = U.dev(w), U.dev(b)
    out = torch.full((G, B, C, T), float('nan'), device=U.DEV)
    if packed:
        nb = lib.vc_gated_conv1d_glu_image_bytes(24, 15, 2 * C)
        img = torch.zeros((G, nb), dtype=torch.uint8, device=U.DEV)
        check(lib.vc_gated_conv1d_glu_prepack(ptr(w_d), w_d[0].numel(), ptr(img), nb, G, 24, 15, 2 * C, U.stream()))
        check(lib.vc_gated_conv1d_glu_fwd_packed(ptr(x_d), x_d[0].numel(), ptr(img), nb, ptr(b_d), 2 * C, ptr(out), out[0].numel(), G, B, 24, T, 2 * C, 15, U.stream()))
    else:
        check(lib.vc_gated_conv1d_glu_fwd(ptr(x_d), x_d[0].numel(), ptr(w_d), w_d[0].numel(), ptr(b_d), 2 * C, ptr(out), out[0].numel(), G, B, 24, T, 2 * C, 15, U.stream()))
    o = out.cpu().numpy()
    check(lib.vc_set_math_mode(0))
    err = np.abs(o - ref)
    bad = ~(err <= 1e-2)
    rel = U.rel_err(np.nan_to_num(o, nan=1e9), ref)
    if rel > 2e-4:
        print('FAIL round', rnd, B, T, mode, packed, 'rel', rel, 'bad frac', bad.mean(), 'nan', np.isnan(o).sum(), flush=True)
        if T % 32 == 0:
            e = bad.reshape(G, B, 4, 32, T // 32, 32)
            print('  by g,b:', bad.reshape(G, B, -1).mean(-1).round(2).tolist())
            print('  by slab:', e.mean((0, 1, 3, 4, 5)).round(3).tolist())
            print('  by chan-in-slab:', e.mean((0, 1, 2, 4, 5)).round(2).tolist())
            print('  by 32px block (first 32):', e.mean((0, 1, 2, 3, 5)).round(2).tolist()[:32])
            print('  by px-in-block:', e.mean((0, 1, 2, 3, 4)).round(2).tolist())
        return 1
    return 0
nfail = 0
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    for B, T in [(2, 128), (256, 128), (3, 132), (8, 2048), (1, 4), (5, 260)]:
        for mode in ('fp32', 'bf16'):
            for packed in (False, True):
                nfail += run(B, T, mode, packed, B * 7 + T, rnd)
print('rounds done, failures:', nfail)
