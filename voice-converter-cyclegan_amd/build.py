"""Builds libvcgan_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

No torch.utils.cpp_extension (it would hipify); plain `hipcc -shared -fPIC`.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(CSRC, 'libvcgan_hip.so')
SOURCES = ['vc_kernels.hip', 'vc_api.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++20', '-munsafe-fp-atomics', '-shared', '-fPIC',
         '-Wall', '-Wextra', '-Wno-unused-parameter']


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = SOURCES + ['vc_internal.h', os.path.join('..', '..', 'include', 'vc_cyclegan.h')]
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc] + FLAGS + os.environ.get('VC_EXTRA_FLAGS', '').split() + ['-o', LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print('[vcgan] ' + ' '.join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == '__main__':
    build(force=True)
