"""Bitwise stress of the named kernel: the launch is deterministic, so every repetition must equal the first one bit for bit.
python tools/stress_glu.py [launches per case] [vc_tune_glu flags]   -- prints the blocks that differ when one does."""
import os, sys
import torch
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge; ge.build()
from voice_converter_cyclegan_amd._lib import lib, ptr, check, vp
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = 'cuda:0'
s = vp(torch.cuda.current_stream().cuda_stream)
lib.vc_tune_glu(flags)
tot_bad = 0
for math in ('bf16', 'fp32'):
    for (G, B, T) in (((1, 1024, 128), (1, 512, 128), (2, 512, 128), (1, 2048, 128)) if os.environ.get('STRESS_BIG') else ((2, 256, 128), (1, 256, 128), (2, 8, 2048), (1, 1024, 128))):
        for packed in (False, True):
            if math == 'fp32' and B * T > 40000: continue
            torch.manual_seed(1)
            x = torch.randn((G, B, 24, T), device=dev); w = torch.randn((G, 15, 24, 256), device=dev) * 0.05; b = torch.randn((G, 256), device=dev) * 0.1
            y0 = torch.empty((G, B, 128, T), device=dev); y = torch.empty_like(y0)
            check(lib.vc_set_math_mode({'fp32': 0, 'bf16': 1}[math]))
            nb = lib.vc_gated_conv1d_glu_image_bytes(24, 15, 256)
            img = torch.zeros((G, nb), dtype=torch.uint8, device=dev)
            check(lib.vc_gated_conv1d_glu_prepack(ptr(w), w[0].numel(), ptr(img), nb, G, 24, 15, 256, s))
            def call(out):
                if packed: check(lib.vc_gated_conv1d_glu_fwd_packed(ptr(x), x[0].numel(), ptr(img), nb, ptr(b), 256, ptr(out), out[0].numel(), G, B, 24, T, 256, 15, s))
                else: check(lib.vc_gated_conv1d_glu_fwd(ptr(x), x[0].numel(), ptr(w), w[0].numel(), ptr(b), 256, ptr(out), out[0].numel(), G, B, 24, T, 256, 15, s))
            call(y0); torch.cuda.synchronize()
            n = N if math == 'bf16' else N // 4
            bad = 0
            for i in range(n):
                y.fill_(float('nan'))
                call(y)
                neq = (y != y0)
                c = int(neq.sum().item())
                if c:
                    bad += 1
                    if bad <= 6:
                        # describe this failure: blocks (g, b, slab of 32 ch, 32-px block) that differ; inside the first one: which channels / pixels
                        e = neq.reshape(G, B, 4, 32, T // 32, 32)
                        blk = e.any(dim=5).any(dim=3)
                        idx = blk.nonzero().tolist()
                        g0, b0, s0, p0 = idx[0]
                        ch = e[g0, b0, s0, :, p0, :].any(dim=1).nonzero().flatten().tolist()
                        px = e[g0, b0, s0, :, p0, :].any(dim=0).nonzero().flatten().tolist()
                        d = (y - y0)[neq]
                        print('   launch %d: %d elements differ (nan %d, max |diff| %.3g) in %d (g, b, slab, px-block) blocks: %s ; first block: channels %s pixels %s' % (
                            i, c, int(torch.isnan(y).sum().item()), float(d[~torch.isnan(d)].abs().max().item()) if (~torch.isnan(d)).any() else float('nan'),
                            len(idx), idx[:16], ch, px), flush=True)
            tot_bad += bad
            print('%s G=%d B=%d T=%d %s: %d launches, %d differ from the first' % (math, G, B, T, 'packed' if packed else 'plain', n, bad), flush=True)
check(lib.vc_set_math_mode(0)); lib.vc_tune_glu(0)
print('total differing launches:', tot_bad)
