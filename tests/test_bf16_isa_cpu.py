"""CPU: the generated ISA of conv_bf16_kernel (csrc/vc_conv_bf16.hip; reference module.py:32-40, 54-62 in bf16 mode) keeps the promise
its ring of LDS stages relies on.  Operands travel HBM -> LDS with `buffer_load_dwordx4 ... lds` copies that no register and no
compiler dependency tracks; a chunk is declared landed by a COUNTED `s_waitcnt vmcnt((NST - 2) * GL)` at the top of the K-loop
iteration, GL = 4 + 2 (BM / 64) copies per chunk and wave.  gfx950 has ONE in-order counter for every vector-memory operation, so the
count is only right if an iteration of the K loop issues exactly GL LDS-direct copies and NOTHING else that counts (a spill, a
hoisted table load, a merged or split copy would declare a chunk landed too early -- garbage operands, silently).  This test reads
the assembly hipcc produces for gfx950 and checks, for every built (BM, NST, ROLE): one K loop with one barrier, GL copies, the
counted wait in front of the barrier, 4 * 2 * (BM / 64) MFMAs, no other vector-memory instruction, no scratch.  The wave-specialised
form (SPEC: the configurations with one workgroup per CU) has TWO loops that meet at the same barrier: the copying waves' (GL copies, the counted wait, nothing else
that counts) and the multiplying waves' (the MFMAs, no vector-memory instruction at all)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'voice-converter-cyclegan_amd', 'csrc', 'vc_conv_bf16.hip')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
CONFIGS = [(bm, nst, role) for bm, nst in ((256, 3), (128, 2), (128, 4), (64, 2), (64, 4)) for role in (0, 1)]


@pytest.fixture(scope='module')
def asm(tmp_path_factory):
    if shutil.which(HIPCC) is None and not os.path.exists(HIPCC):
        pytest.skip('hipcc not available')
    out = str(tmp_path_factory.mktemp('bf16') / 'conv_bf16.s')
    from importlib.util import spec_from_file_location, module_from_spec
    spec = spec_from_file_location('_vcb', os.path.join(ROOT, 'voice-converter-cyclegan_amd', 'build.py'))
    b = module_from_spec(spec); spec.loader.exec_module(b)
    subprocess.check_call([HIPCC] + b.FLAGS + ['-S', '--cuda-device-only', SRC, '-o', out], cwd=os.path.dirname(SRC),
                          stderr=subprocess.DEVNULL)
    return open(out).read()


def _kernel(text, bm, nst, role, spec=False):
    lines = text.split('\n')
    tag = '_Z16conv_bf16_kernelILi%dELi%dELi%dELb%dEEv' % (bm, nst, role, int(spec))
    starts = [i for i, l in enumerate(lines) if l.startswith(tag) and ':' in l.split(';')[0]]
    assert len(starts) == 1, (tag, len(starts))
    body = []
    for l in lines[starts[0] + 1:]:
        if l.startswith('.Lfunc_end'):
            break
        body.append(l)
    return body


def _instrs(block):
    return [l.strip() for l in block if l.startswith('\t') and not l.strip().startswith(('.', ';'))]


def _loops(body):
    loops, cur = {}, None
    for l in body:
        if re.match(r'^(\.LBB\d+_\d+:|; %bb\.\d+:)', l):
            h = re.search(r'in Loop: Header=(BB\d+_\d+)', l)
            cur = h.group(1) if h else (l.split(':')[0].lstrip('.L') if 'Loop Header' in l else None)
        if cur is not None:
            loops.setdefault(cur, []).append(l)
    return loops


@pytest.mark.parametrize('bm,nst,role', CONFIGS)
def test_k_loop_issues_the_counted_copies(asm, bm, nst, role):
    body = _kernel(asm, bm, nst, role)
    gl = 4 + 2 * (bm // 64)
    loops = _loops(body)
    kloop = [v for v in loops.values() if any(x.startswith('v_mfma') for x in _instrs(v))]
    assert len(kloop) == 1, (bm, nst, role, list(loops))
    ins = _instrs(kloop[0])
    copies = [x for x in ins if re.match(r'buffer_load_dwordx4 .*\blds\b', x)]
    other_vm = [x for x in ins if re.match(r'(buffer_|global_|flat_|scratch_)', x) and x not in copies]
    waits = [x for x in ins if 'vmcnt' in x]
    mfma = [x for x in ins if x.startswith('v_mfma_f32_32x32x16_bf16')]
    assert len(copies) == gl, (len(copies), gl)
    assert not other_vm, other_vm[:3]
    assert waits == ['s_waitcnt vmcnt(%d)' % ((nst - 2) * gl)], waits
    assert len(mfma) == 4 * 2 * (bm // 64), len(mfma)
    assert len([x for x in ins if x == 's_barrier']) == 1
    # program order: the counted wait, the barrier, then reads / MFMAs / refill copies
    w, b = ins.index(waits[0]), ins.index('s_barrier')
    first_copy, first_mfma = ins.index(copies[0]), ins.index(mfma[0])
    assert w < b < first_mfma < first_copy, (w, b, first_mfma, first_copy)
    # the copies are spread among the MFMAs of the four K steps (an LDS-direct copy is ~60 cycles of issue: GL of them in a row at
    # the top of the iteration left the matrix pipe idle for 0.5 - 0.7 us per chunk in round 3): never more than a quarter (+ 1)
    # back to back, and MFMAs on both sides of every run
    seq = ''.join('M' if x.startswith('v_mfma') else 'C' for x in ins if x.startswith('v_mfma') or x in copies)
    runs = [len(r) for r in re.findall(r'C+', seq)]
    assert max(runs) <= gl // 4 + 1, (seq, runs)
    assert seq[0] == 'M' and len(runs) >= 4, seq

SPEC_CONFIGS = [(256, 3), (128, 4), (64, 4)]           # the one-workgroup-per-CU configurations


@pytest.mark.parametrize('bm,nst', SPEC_CONFIGS)
@pytest.mark.parametrize('role', (0, 1))
def test_wave_specialised_loops(asm, role, bm, nst):
    gl = 4 + 2 * (bm // 64)
    loops = [_instrs(v) for v in _loops(_kernel(asm, bm, nst, role, spec=True)).values()]
    loops = [v for v in loops if 's_barrier' in v]
    assert len(loops) == 2, len(loops)
    mult = [v for v in loops if any(x.startswith('v_mfma') for x in v)]
    copy = [v for v in loops if not any(x.startswith('v_mfma') for x in v)]
    assert len(mult) == 1 and len(copy) == 1
    mult, copy = mult[0], copy[0]
    vm = lambda ins: [x for x in ins if re.match(r'(buffer_|global_|flat_|scratch_)', x)]
    # multiplying waves: the chunk's MFMAs between two barriers, fragments by ds_read_b128, nothing the vector-memory counter sees
    assert len([x for x in mult if x.startswith('v_mfma_f32_32x32x16_bf16')]) == 4 * 2 * (bm // 64)
    assert not vm(mult) and not [x for x in mult if 'vmcnt' in x], (vm(mult)[:3], [x for x in mult if 'vmcnt' in x])
    assert mult.count('s_barrier') == 1
    assert len([x for x in mult if x.startswith('ds_read_b128')]) == 4 * (2 + bm // 64)
    # copying waves: exactly the chunk's GL pieces per trip, retired by the counted wait in front of the barrier
    copies = [x for x in copy if re.match(r'buffer_load_dwordx4 .*\blds\b', x)]
    assert len(copies) == gl and vm(copy) == copies, (len(copies), [x for x in vm(copy) if x not in copies][:3])
    assert [x for x in copy if 'vmcnt' in x] == ['s_waitcnt vmcnt(%d)' % ((nst - 2) * gl)]
    assert copy.count('s_barrier') == 1
    w, b, c0 = copy.index('s_waitcnt vmcnt(%d)' % ((nst - 2) * gl)), copy.index('s_barrier'), copy.index(copies[0])
    assert w < b < c0, (w, b, c0)


def test_no_scratch(asm):
    seen = {}
    for blk in re.split(r'\n  - \.agpr_count', asm)[1:]:
        n = re.search(r'\.name:\s+(\S+)', blk); s = re.search(r'\.private_segment_fixed_size:\s+(\d+)', blk)
        if n and s and 'conv_bf16_kernel' in n.group(1):
            seen[n.group(1)] = int(s.group(1))
    assert len(seen) == len(CONFIGS) + 2 * len(SPEC_CONFIGS), seen          # + the wave-specialised forms
    assert all(v == 0 for v in seen.values()), seen
