import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge; ge.build()
from tests import gpu_util as U
from voice_converter_cyclegan_amd._lib import lib, ptr, check
from oracle import cyclegan_oracle as O
def run(B, T, mode, packed, seed):
    rng = np.random.default_rng(seed)
    G, C = 2, 128
    x = rng.standard_normal((G, B, T, 24)).astype(np.float32)
    w = (rng.standard_normal((G, 15, 24, 2 * C)) / np.sqrt(360)).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, (G, 2 * C)).astype(np.float32)
    check(lib.vc_set_math_mode({'fp32': 0, 'bf16': 1}[mode]))
    O.set_operand_rounding('bf16' if mode == 'bf16' else None)
    ref = []
    for g in range(G):
        v, _ = O.conv1d_fwd(x[g], w[g][:, :, :C], b[g][:C], 1)
        gt, _ = O.conv1d_fwd(x[g], w[g][:, :, C:], b[g][C:], 1)
        ref.append(np.transpose(O.glu_fwd(v, gt)[0], (0, 2, 1)))
    ref = np.stack(ref)
    x_d = U.dev(np.ascontiguousarray(np.transpose(x, (0, 1, 3, 2)))); w_d, b_d = U.dev(w), U.dev(b)
    for rep in range(3):
        out = torch.full((G, B, C, T), float('nan'), device=U.DEV)
        if packed:
            nb = lib.vc_gated_conv1d_glu_image_bytes(24, 15, 2 * C)
            img = torch.zeros((G, nb), dtype=torch.uint8, device=U.DEV)
            check(lib.vc_gated_conv1d_glu_prepack(ptr(w_d), w_d[0].numel(), ptr(img), nb, G, 24, 15, 2 * C, U.stream()))
            check(lib.vc_gated_conv1d_glu_fwd_packed(ptr(x_d), x_d[0].numel(), ptr(img), nb, ptr(b_d), 2 * C, ptr(out), out[0].numel(), G, B, 24, T, 2 * C, 15, U.stream()))
        else:
            check(lib.vc_gated_conv1d_glu_fwd(ptr(x_d), x_d[0].numel(), ptr(w_d), w_d[0].numel(), ptr(b_d), 2 * C, ptr(out), out[0].numel(), G, B, 24, T, 2 * C, 15, U.stream()))
        o = out.cpu().numpy()
        err = np.abs(o - ref)
        bad = err > 1e-2
        print(B, T, mode, packed, 'rep', rep, 'rel', U.rel_err(o, ref), 'bad frac', bad.mean(), 'nan', np.isnan(o).sum())
        if bad.any():
            # pattern: per (g, b), per slab (32 ch), per 128-px tile, per 32-px block
            e = bad.reshape(G, B, 4, 32, T // 32, 32)
            print('  by g,b:', bad.reshape(G, B, -1).mean(-1).round(2).tolist())
            print('  by slab:', e.mean((0, 1, 3, 4, 5)).round(3).tolist())
            print('  by chan-in-slab:', e.mean((0, 1, 2, 4, 5)).round(2).tolist())
            print('  by 32px block (first 32):', e.mean((0, 1, 2, 3, 5)).round(2).tolist()[:32])
    O.set_operand_rounding(None); check(lib.vc_set_math_mode(0))
run(8, 2048, 'bf16', True, 8 * 7 + 2048)
run(8, 2048, 'bf16', False, 8 * 7 + 2048)
run(8, 2048, 'fp32', True, 8 * 7 + 2048)
run(16, 1024, 'bf16', True, 5)
run(64, 128, 'bf16', True, 6)
