"""CPU: the generated ISA of gated_conv1d_glu_kernel (csrc/vc_conv_glu.hip; reference module.py:161-163) keeps the promises its
hand-counted waits rely on.  The kernel issues the frames of the next-but-one tile with inline-asm buffer loads into registers
the compiler knows nothing about until a counted `s_waitcnt vmcnt(8 + 2 NST)` one trip later; that is only correct if
  * every trip issues exactly 8 such loads and exactly NST output stores (8 inference, 24 training: dword stores of two full
    128-byte lines each) -- no merged, split or branch-skipped store --, with no other vector-memory operation in between,
  * nothing reads or overwrites the destination registers between the loads and the wait that retires them,
  * no register is spilled (scratch would add vector-memory operations of its own) and the loop holds no vmcnt(0),
  * the two bodies of a kernel (K set 0 / 1: a wave-uniform branch at entry picks one) execute the same number of workgroup
    barriers per trip -- waves of both sets meet at every one of them.
A compiler or flag change that breaks one of these would stage garbage frames silently; this test reads the
assembly hipcc produces for gfx950 and fails instead."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'voice-converter-cyclegan_amd', 'csrc', 'vc_conv_glu.hip')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


@pytest.fixture(scope='module')
def asm(tmp_path_factory):
    if shutil.which(HIPCC) is None and not os.path.exists(HIPCC):
        pytest.skip('hipcc not available')
    out = str(tmp_path_factory.mktemp('glu') / 'glu.s')
    from importlib.util import spec_from_file_location, module_from_spec
    spec = spec_from_file_location('_vcb', os.path.join(ROOT, 'voice-converter-cyclegan_amd', 'build.py'))
    b = module_from_spec(spec); spec.loader.exec_module(b)
    subprocess.check_call([HIPCC] + b.FLAGS + ['-S', '--cuda-device-only', SRC, '-o', out], cwd=os.path.dirname(SRC),
                          stderr=subprocess.DEVNULL)
    return open(out).read()


def _kernels(text):
    lines = text.split('\n')
    starts = [i for i, l in enumerate(lines) if re.match(r'^_Z23gated_conv1d_glu_kernel\w+:', l)]
    for k, a in enumerate(starts):
        name = lines[a].split(':')[0]
        body = []
        for l in lines[a + 1:]:
            if l.startswith('.Lfunc_end'):
                break
            body.append(l)
        yield name, body


def _instrs(body):
    return [l.strip() for l in body if l.startswith('\t') and not l.strip().startswith(('.', ';'))]


def test_no_scratch_and_register_budget(asm):
    seen = {}
    for blk in re.split(r'\n  - \.agpr_count', asm)[1:]:
        n = re.search(r'\.name:\s+(\S+)', blk); s = re.search(r'\.private_segment_fixed_size:\s+(\d+)', blk)
        v = re.search(r'\.vgpr_count:\s+(\d+)', blk)
        if n and s and 'gated_conv1d_glu_kernel' in n.group(1):
            seen[n.group(1)] = (int(s.group(1)), int(v.group(1)) if v else 0)
    assert len(seen) == 12, seen          # fp32, bf16 and IEEE-half operands x (GLU only | + raw outputs) x (filters converted in place | pre-packed)
    assert all(v[0] == 0 for v in seen.values()), 'scratch in use: %s' % seen
    # two waves per SIMD: the unified register file gives each 256 registers (VGPR + AGPR)
    assert all(0 < v[1] <= 256 for v in seen.values()), seen
    assert 'scratch_' not in asm


@pytest.mark.parametrize('pre', [False, True])
@pytest.mark.parametrize('bf16,raw,half', [(True, False, False), (True, True, False), (False, False, False), (False, True, False),
                                           (True, False, True), (True, True, True)])
def test_trip_issues_the_counted_operations(asm, bf16, raw, pre, half):
    """half: the 16-bit form with IEEE-half operands (math mode 3) -- the bf16 kernel with another conversion and another MFMA, same schedule"""
    tag = 'ILi24ELi15ELb%dELb%dELb%dELb%dEE' % (int(bf16), int(raw), int(pre), int(half))
    body = [b for n, b in _kernels(asm) if tag in n]
    assert len(body) == 1
    body = body[0]
    nst = 24 if raw else 8
    want = 's_waitcnt vmcnt(%d)' % (8 + 2 * nst)
    # The kernel holds two tile loops (K set 0 / 1), each two trips long (register sets alternate), each trip ending in the
    # workgroup barrier(s).  Block placement is the compiler's business: keep the lines of blocks LLVM annotates as belonging
    # to a loop that holds a barrier, then cut each loop's stream at its barriers (fp32: two per trip).
    loops = {}
    cur = None
    for l in body:
        m = re.match(r'^(\.LBB\d+_\d+:|; %bb\.\d+:)', l)
        if m:
            h = re.search(r'in Loop: Header=(BB\d+_\d+)', l)
            cur = h.group(1) if h else (l.split(':')[0].lstrip('.L') if 'Loop Header' in l else None)
        if cur is not None:
            loops.setdefault(cur, []).append(l)
    tile_loops = [v for v in loops.values() if any('s_barrier' in l for l in v)]
    assert len(tile_loops) == 2, (tag, list(loops))
    bars_per_trip = 1 if bf16 else 2
    mfmas = set()
    for stream in tile_loops:
        bars = [i for i, l in enumerate(stream) if 's_barrier' in l]
        assert len(bars) == 2 * bars_per_trip, (tag, len(bars))
        ends = bars[bars_per_trip - 1::bars_per_trip]
        cuts = [-1] + ends
        trips_l = [stream[lo + 1:hi + 1] for lo, hi in zip(cuts[:-1], cuts[1:])]
        assert len(trips_l) == 2 and all(any(want in l for l in t) for t in trips_l), (tag, len(trips_l))
        body2 = [l for t in trips_l for l in t]
        n0 = len(trips_l[0])
        trips = [(0, n0), (n0, len(body2))]
        dests = []
        for lo, hi in trips:
            ins = _instrs(body2[lo:hi])
            loads = [x for x in ins if x.startswith('buffer_load_dword ')]
            stores = [x for x in ins if x.startswith('buffer_store_dword ')]
            other_vm = [x for x in ins if re.match(r'(buffer_|global_|flat_|scratch_)', x) and x not in loads and x not in stores]
            waits = [x for x in ins if 'vmcnt' in x]
            assert len(loads) == 8, (len(loads), tag)
            assert len(stores) == nst, (len(stores), tag)
            assert not other_vm, other_vm[:3]
            assert waits == [want], waits
            mfmas.add(len([x for x in ins if x.startswith('v_mfma')]))
            # program order inside the trip: loads of the next-but-one tile, then the previous tile's stores, then the wait
            order = [x.split()[0] if 'vmcnt' not in x else 'wait' for x in ins if x in loads or x in stores or 'vmcnt' in x]
            assert order.index('wait') == len(order) - 1, 'a vector-memory operation behind the counted wait of its trip'
            dests.append({re.match(r'buffer_load_dword (v\d+),', x).group(1) for x in loads})
        assert not (dests[0] & dests[1]), 'the two register sets overlap'
        # a trip's load destinations are retired by the OTHER trip's wait: until then no instruction may touch them
        for k, (lo, hi) in enumerate(trips):
            mine, others = dests[k], dests[1 - k]
            ins = _instrs(body2[lo:hi])
            w = [i for i, x in enumerate(ins) if 'vmcnt' in x][0]
            last_load = max(i for i, x in enumerate(ins) if x.startswith('buffer_load_dword '))
            for i, x in enumerate(ins):
                regs = set(re.findall(r'\bv(\d+)\b', x))
                for a, b in re.findall(r'v\[(\d+):(\d+)\]', x):
                    regs |= {str(r) for r in range(int(a), int(b) + 1)}
                regs = {'v' + r for r in regs}
                if i < w and not x.startswith('buffer_load_dword '):
                    assert not (regs & others), 'trip %d touches frames in flight before its wait: %s' % (k, x)
                if i > last_load:
                    assert not (regs & mine), 'trip %d touches its own loads before the next trip\'s wait: %s' % (k, x)
    # the K axis is split 12 / 11 steps (bf16: two MFMAs per step) or 23 / 22 (fp32: eight per step) between the two bodies
    assert mfmas == ({24, 22} if bf16 else {184, 176}), mfmas
    mf = {x.split()[0] for x in _instrs(body) if x.startswith('v_mfma')}
    assert mf == ({'v_mfma_f32_32x32x16_f16'} if half else {'v_mfma_f32_32x32x16_bf16'} if bf16 else {'v_mfma_f32_32x32x2_f32'}), mf


def _vregs(x):
    r = set(re.findall(r'\bv(\d+)\b', x))
    for a, b in re.findall(r'v\[(\d+):(\d+)\]', x):
        r |= {str(k) for k in range(int(a), int(b) + 1)}
    return {'v' + k for k in r}


def test_nothing_is_in_flight_at_the_loop_entry_and_nothing_touches_frames_in_flight(asm):
    """Round 5 found the hole in the checks above: they police the tile loop's body, and the compiler placed register copies
    (v_mov of the second frame set, phi copies of the loop's entry) BETWEEN the prologue's inline-asm loads of tile 1 and the first
    trip's wait -- a copy of a register whose load has not landed is a copy of garbage: 0.1 - 0.4 % of the launches of the un-packed
    bf16 kernel at B >= 512 computed their second tile from it (tools/stress_glu.py; tests/test_ops_gpu.py::
    test_gated_conv1d_glu_is_bitwise_repeatable).  The prologue now retires BOTH tiles' loads with one wait before the loop.
    Checked here, for all 12 instantiations and both bodies: every group of 8 inline-asm frame loads is followed -- in program text --
    by the inline-asm wait that retires it with no instruction in between that reads or writes one of its destination registers;
    and each body's first two groups (the prologue's) reach their wait BEFORE the first loop header behind them."""
    n_kernels = 0
    for name, body in _kernels(asm):
        n_kernels += 1
        idx = [i for i, l in enumerate(body) if l.strip().startswith('buffer_load_dword ') and 'ASMSTART' in body[i - 1]]
        assert len(idx) == 64, (name, len(idx))             # 2 bodies x (2 prologue + 2 loop) groups of 8
        groups = [idx[k:k + 8] for k in range(0, len(idx), 8)]
        loops = [i for i, l in enumerate(body) if 'Loop Header' in l]
        for gi, g in enumerate(groups):
            dests = {re.match(r'\s*buffer_load_dword (v\d+),', body[i]).group(1) for i in g}
            assert len(dests) == 8, (name, g[0])
            j = g[-1] + 1
            while not (body[j].strip().startswith('s_waitcnt vmcnt') and 'ASMSTART' in body[j - 1]):
                l = body[j].strip()
                if l and not l.startswith((';', '.')) and not l.startswith('buffer_load_dword'):
                    assert not (_vregs(l) & dests), '%s: frames in flight (loads at line %d) touched by: %s' % (name[-30:], g[0], l)
                j += 1
            if gi % 4 < 2:                                   # the prologue's two groups of this body
                nxt = [lh for lh in loops if lh > g[0]]
                assert nxt and j < nxt[0], '%s: the prologue\'s loads (line %d) are still in flight at the loop entry (wait at %d, loop at %s)' % (name[-30:], g[0], j, nxt[:1])
    assert n_kernels == 12
