"""CPU: the C-ABI library builds for gfx950, loads without a GPU, exports every symbol the
public header declares, and its parameter layout speaks the reference's TF variable names."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions(names=('vc_cyclegan.h', 'vc_cyclegan_testing.h')):
    out = set()
    for name in names:
        src = open(os.path.join(ROOT, 'include', name)).read()
        src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
        out |= set(re.findall(r'\b(vc_[a-z0-9_]+)\s*\(', src))
    return sorted(out)


def test_product_header_carries_no_test_hooks():
    """vc_tune_* / vc_prof_* live in include/vc_cyclegan_testing.h; the product path (engine / model / module) calls none."""
    prod = _header_functions(('vc_cyclegan.h',))
    assert not [n for n in prod if n.startswith(('vc_tune_', 'vc_prof_', 'vc_testing_'))]
    pkg = os.path.join(ROOT, 'voice-converter-cyclegan_amd')
    for f in ('model.py', 'module.py', 'data.py', 'tfio.py', 'layout.py'):
        src = open(os.path.join(pkg, f)).read()
        assert 'vc_tune_' not in src and 'vc_testing_' not in src and 'vc_prof_' not in src, f


def test_header_symbols_exported_and_bound(built_lib):
    names = _header_functions()
    assert len(names) >= 30
    bound = {n for n, _, _ in built_lib.SIGNATURES}
    for n in names:
        assert hasattr(built_lib.lib, n), 'symbol %s declared in include/*.h is not exported' % n
        assert n in bound, 'symbol %s has no ctypes signature in _lib.SIGNATURES' % n
    assert bound <= set(names), 'python binds symbols the header does not declare: %s' % (bound - set(names))
    assert built_lib.lib.vc_version() == built_lib.ABI_VERSION == 116


def test_library_is_stamped_with_the_hash_of_its_sources(built_lib, tmp_path, monkeypatch):
    """freshness by content, not by mtime.  The library reports the hash of the sources + flags it was built
    from; it equals the hash of the tree (else _lib would have refused to load); the hash moves with any source byte or flag;
    and a library whose id differs from the tree's is refused at import."""
    import importlib.util
    import subprocess
    import sys
    spec = importlib.util.spec_from_file_location('_vcb_id', os.path.join(ROOT, 'voice-converter-cyclegan_amd', 'build.py'))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    bid = built_lib.lib.vc_build_id().decode()
    assert re.fullmatch(r'[0-9a-f]{16}', bid), bid
    assert bid == b.build_id() == built_lib.BUILD_ID
    assert not b.needs_build()
    assert b.build_id(['-DX']) != bid                      # flags are part of the identity
    # an older-but-different source (mtime BEFORE the library's) still invalidates: emulate by hashing a modified copy
    src = os.path.join(b.CSRC, b.SOURCES[0])
    orig = open(src, 'rb').read()
    monkeypatch.setattr(b, 'CSRC', str(tmp_path))
    for f in b.SOURCES + b.HEADERS:
        dst = os.path.join(str(tmp_path), f)
        os.makedirs(os.path.dirname(os.path.abspath(dst)), exist_ok=True)
        with open(os.path.abspath(dst), 'wb') as fh:
            fh.write(open(os.path.join(os.path.dirname(src), f), 'rb').read())
    assert b.build_id() == bid
    with open(os.path.join(str(tmp_path), b.SOURCES[0]), 'wb') as fh:
        fh.write(orig + b'\n// touched\n')
    os.utime(os.path.join(str(tmp_path), b.SOURCES[0]), (1, 1))      # far older than the library
    assert b.build_id() != bid
    # ... and the loader refuses a library whose id is not the tree's (a sub-process: the module is already imported here)
    # (VC_EXTRA_FLAGS is part of the identity the loader expects: with it set, the tree's id is no longer the library's)
    code = ('import sys, os; sys.path.insert(0, %r); import vcgan_pkg\n'
            'os.environ["VC_EXTRA_FLAGS"] = "-DSOMETHING_ELSE"\n'
            'try:\n'
            '    from voice_converter_cyclegan_amd import _lib\n'
            '    print("LOADED")\n'
            'except ImportError as e:\n'
            '    print("REFUSED", e)\n') % (ROOT,)
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, cwd=ROOT).stdout
    assert 'REFUSED' in out and 'build id' in out, out


def test_library_is_gfx950_only(built_lib):
    """no multi-arch / CUDA dual paths: the fat binary carries exactly one device target."""
    blob = open(built_lib.LIB_PATH, 'rb').read()
    targets = set(re.findall(rb'amdgcn-amd-amdhsa--(gfx[0-9a-f]+)', blob))
    assert targets == {b'gfx950'}, targets


def _code_objects(path):
    """the gfx950 ELF images inside the library's clang offload bundles (one per translation unit with kernels)"""
    import struct
    blob = open(path, 'rb').read()
    magic, pos = b'__CLANG_OFFLOAD_BUNDLE__', 0
    while True:
        i = blob.find(magic, pos)
        if i < 0:
            return
        n, = struct.unpack_from('<Q', blob, i + 24)
        p = i + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from('<QQQ', blob, p)
            triple = blob[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if 'gfx' in triple and size:
                yield triple, blob[i + off:i + off + size]
        pos = i + 24


def test_no_kernel_uses_scratch_or_spills_vector_registers(built_lib, tmp_path):
    """Round 4 found 12 bytes of private memory per lane in all 39 igemm instantiations (a reference swap in the pixel decode) --
    a store / load round trip through the memory system in every prologue and epilogue.  The kernel metadata of the
    built library is the cheap place to keep that from coming back: every kernel has private_segment_fixed_size 0 and no VGPR
    spills (SGPR spills into spare VGPR lanes are allowed: the igemm epilogue pins more scalars than there are SGPRs)."""
    import subprocess
    readelf = '/opt/rocm/lib/llvm/bin/llvm-readelf'
    if not os.path.exists(readelf):
        pytest.skip('no llvm-readelf')
    kernels, bad = 0, []
    for k, (triple, image) in enumerate(_code_objects(built_lib.LIB_PATH)):
        assert triple.endswith('gfx950'), triple
        fn = tmp_path / ('co%d.elf' % k)
        fn.write_bytes(image)
        notes = subprocess.run([readelf, '--notes', str(fn)], capture_output=True, text=True, check=True).stdout
        for rec in notes.split('- .agpr_count:')[1:]:
            name = re.search(r'\.name:\s+(\S+)', rec).group(1)
            priv = int(re.search(r'\.private_segment_fixed_size:\s+(\d+)', rec).group(1))
            vsp = int(re.search(r'\.vgpr_spill_count:\s+(\d+)', rec).group(1))
            kernels += 1
            if priv or vsp:
                bad.append((name, priv, vsp))
    assert kernels >= 200, kernels                  # 238 at the time of writing: every instantiation is looked at
    assert not bad, bad


def test_layout_matches_oracle_names_and_counts(built_lib):
    from voice_converter_cyclegan_amd.layout import ParamLayout, GENERATOR, DISCRIMINATOR
    from oracle import cyclegan_oracle as O
    G, D = ParamLayout(GENERATOR, 24), ParamLayout(DISCRIMINATOR, 24)
    assert (G.n_params, len(G.recs)) == (38055704, 110)
    assert (D.n_params, len(D.recs)) == (21837825, 30)
    assert G.n_floats % 4 == 0 and D.n_floats % 4 == 0
    assert [r['name'] for r in G.recs and sorted(G.recs, key=lambda r: r['name'])] == \
        sorted(n for n, _ in O.generator_layout(24))
    assert {r['name']: r['shape'] for r in G.recs} == dict(O.generator_layout(24))
    assert {r['name']: r['shape'] for r in D.recs} == dict(O.discriminator_layout())
    # 36-dim MCEP (BASELINE config 5): only h1 / o1 change
    G36 = ParamLayout(GENERATOR, 36)
    assert {r['name']: r['shape'] for r in G36.recs} == dict(O.generator_layout(36))
    # pack/unpack round trip with value and gate kernels interleaved in one stored tensor
    P = O.init_params(seed=1, randomize_affine=True)
    flat = G.pack(P, 'generator_B2A')
    back = G.unpack(flat, 'generator_B2A')
    assert all(np.array_equal(back[k], P[k]) for k in back)
    # the records tile the block: every float belongs to exactly one variable (no overlap), the rest is alignment padding
    for L, scope in ((G, 'generator_A2B'), (D, 'discriminator_A')):
        ones = {scope + '/' + x['name']: np.ones(x['shape'], np.float32) for x in L.recs}
        flat1 = L.pack(ones, scope)
        twos = {k: 2 * v for k, v in ones.items()}
        assert float(flat1.sum(dtype=np.float64)) == L.n_params and int((flat1 == 0).sum()) == L.n_floats - L.n_params
        assert float(L.pack(twos, scope).sum(dtype=np.float64)) == 2 * L.n_params and L.n_floats - L.n_params < 64
    r = {x['name']: x for x in G.recs}
    a, b = r['residual1d_block2_h1_conv/kernel'], r['residual1d_block2_h1_gates/kernel']
    assert b['offset'] == a['offset'] + 1024 and a['ld'] == b['ld'] == 2048


def test_argument_errors_without_gpu(built_lib):
    lib = built_lib.lib
    n = built_lib.i64(0)
    assert lib.vc_param_layout(7, 24, None, 0, C.byref(n)) == -1
    assert b'kind' in lib.vc_last_error()
    assert lib.vc_param_layout(0, 25, None, 0, C.byref(n)) == -1          # num_features % 4
    h = built_lib.vp()
    assert lib.vc_plan_create(24, 130, C.byref(h)) == -1                  # n_frames % 4 (preprocess.py:146-156)
    assert lib.vc_conv2d_fwd(None, 0, None, 0, None, 0, None, 0, 1, 1, 1, 1, 1, 4, 1, 1, 1, 1, None) == -1
    assert lib.vc_adam_tf_update(None, None, None, None, 0, 0.0, 0.5, 0.999, 1e-8, 1.0, None) == -1


def test_engine_fails_loudly_without_gpu(built_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from voice_converter_cyclegan_amd.engine import CycleGANEngine
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        CycleGANEngine(24)


def test_product_path_never_imports_oracle():
    pkg = os.path.join(ROOT, 'voice-converter-cyclegan_amd')
    for f in os.listdir(pkg) + ['../module.py', '../model.py', '../utils.py', '../vcgan_pkg.py']:
        if f.endswith('.py'):
            src = open(os.path.join(pkg, f)).read()
            assert 'oracle' not in src, f
