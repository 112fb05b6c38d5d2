"""Which production step carries the bad 256-byte run in a stacked pixel-octet buffer, and what is in it?  Six steps per trial:
S0 prod(other) S1 prod(lam) S2 prod(other) S3 ctrl(lam) S4 ctrl(lam) S5 prod(lam); truth for the dYq buffers of schedule lam = S3 (= S4)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge; ge.build()
from tests.test_step_gpu import _setup
from voice_converter_cyclegan_amd._lib import lib, check
import ctypes as C
B = 8
O, P, A, Bm, eng = _setup(B, gap=False)
eng.set_math('bf16')
check(lib.vc_tune_conv_bf16_override(128, 4, 1)); check(lib.vc_tune_igemm_override(64, 1, 1)); check(lib.vc_tune_wgrad_override(64, 1, 1)); check(lib.vc_tune_c1_dgrad_gather(1))
A_d, B_d = eng._to_dev(A), eng._to_dev(Bm)
OV = os.environ.get('OVERLAP')
if os.environ.get('LAYER_UPDATES') is not None:
    check(lib.vc_tune_layer_updates(int(os.environ['LAYER_UPDATES'])))

gA = torch.empty((B, 24, 128), device='cuda:0')


def regions(p, Bbwd):
    n = lib.vc_testing_describe_step(p, B, Bbwd, None, 0)
    buf = C.create_string_buffer(int(n))
    lib.vc_testing_describe_step(p, B, Bbwd, buf, n)
    return [(l.split()[0], int(l.split()[1]), int(l.split()[2])) for l in buf.value.decode().splitlines()]


def run(mode, lam):
    check(lib.vc_tune_bf16_pixel_octets(mode))
    if OV is not None:
        check(lib.vc_plan_set_overlap(eng.plan(128), int(OV)))
    p = eng.step_grads(A_d, B_d, 10.0, lam, gen_A=gA); torch.cuda.synchronize()
    return p, eng._ws.clone()


found = 0
for trial in range(80):
    lam = 0.0 if trial % 2 else 5.0
    PM = int(os.environ.get('PROD_MODE', '1'))
    seq = [(PM, 5.0 - lam), (PM, lam), (PM, 5.0 - lam), (0, lam), (0, lam), (PM, lam)]
    S = []
    for m, l in seq:
        p, w = run(m, l); S.append(w)
    regs = [(n, o, ln) for n, o, ln in regions(p, B if lam == 0.0 else 2 * B) if n.startswith('SCRC.step9') and n.endswith('.dYq')]
    Bf = B if lam == 0.0 else 2 * B
    for k in (1, 5):
        for name, off, ln in regs:
            L = int(name.split('.L')[1].split('.')[0])
            a, t = S[k][off:off + ln].view(torch.int32), S[3][off:off + ln].view(torch.int32)
            # only the part both modes write: cycle-hop B samples + first-hop Bf samples
            neq = (a != t)
            nz = torch.nonzero(neq).flatten()
            if nz.numel() == 0 or nz.numel() > 4096:
                continue
            lo, hi = int(nz[0]), int(nz[-1]) + 1
            prev = S[k - 1][off + lo:off + hi].view(torch.int32)
            cur = a[lo:hi]; tru = t[lo:hi]
            if (lo // 4) // {True: 2048, False: 512}[L % 2 == 1] // 4 >= B + Bf:
                continue                      # beyond the samples this schedule writes: leftovers of earlier steps on both sides
            found += 1
            ii = nz.cpu().numpy(); brk = np.nonzero(np.diff(ii) > 1)[0]
            runs = [(int(ii[s_]), int(ii[e_]) + 1) for s_, e_ in zip(np.concatenate([[0], brk + 1]), np.concatenate([brk, [ii.size - 1]]))]
            Co = 2048 if L % 2 == 1 else 512
            print('     runs of differing floats (float index; pixel octet, channel, bf16 pair in the unit): ' +
                  '  '.join('[%d,%d)=po %d ch %d+%d' % (a_, b_, (a_ // 4) // Co, (a_ // 4) % Co, a_ % 4) for a_, b_ in runs[:24]))
            va = S[k][off + lo:off + lo + 8].view(torch.int16).to(torch.int32); vt = S[3][off + lo:off + lo + 8].view(torch.int16).to(torch.int32)
            print('     first 16 bf16 values  got  ', [round(float(x), 6) for x in (va << 16).view(torch.float32)])
            print('                           truth', [round(float(x), 6) for x in (vt << 16).view(torch.float32)])
            print('trial %d lam %g: S%d %s: %d floats differ from the control step, span [%d, %d) (%d bytes), byte offset in the buffer %% 128 = %d'
                  % (trial, lam, k, name, nz.numel(), lo, hi, 4 * (hi - lo), (4 * lo) % 128))
            print('     span equals the PREVIOUS step S%d (%s lam %g) in %d of %d floats; equals truth in %d' % (k - 1, 'prod' if seq[k - 1][0] else 'ctrl', seq[k - 1][1], int((cur == prev).sum()), hi - lo, int((cur == tru).sum())))
            # where: [po][Cout][8 bf16] -> float index = (po * Cout + ch) * 4
            Cout = {True: 2048, False: 512}[L % 2 == 1 and 3 <= L <= 14] if 3 <= L <= 14 else None
            if Cout:
                po, ch = (lo // 4) // Cout, (lo // 4) % Cout
                print('     pixel octet %d (sample %d of the stacked buffer, octet %d in it), channels %d .. %d' % (po, po // 4, po % 4, ch, ch + (hi - lo) // 4 - 1))
print('bad runs found: %d in %d trials' % (found, trial + 1))
