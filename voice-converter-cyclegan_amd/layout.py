"""Flat parameter blocks <-> TensorFlow variable names.

One network's parameters live in one contiguous fp32 block whose layout is
defined by the native library (vc_param_layout).  The reference graph would
create the variables `<scope>/<name>` listed in SURVEY.md section 8a row 13
(module.py:154-159,194-199); this module maps between the two so checkpoints and
the parity tests can speak TF names.
"""
import ctypes as C
import numpy as np

from . import _lib

GENERATOR, DISCRIMINATOR = 0, 1


class ParamLayout:
    def __init__(self, kind, num_features=24):
        n = _lib.i64(0)
        cnt = _lib.lib.vc_param_layout(kind, num_features, None, 0, C.byref(n))
        if cnt <= 0:
            _lib.check(cnt if cnt < 0 else -1, 'vc_param_layout')
        recs = (_lib.ParamRec * cnt)()
        _lib.lib.vc_param_layout(kind, num_features, recs, cnt, C.byref(n))
        self.kind, self.num_features = kind, num_features
        self.n_floats = int(n.value)
        self.recs = [dict(name=r.name.decode(), offset=int(r.offset),
                          shape=tuple(r.shape[i] for i in range(r.ndim)), ld=int(r.ld)) for r in recs]
        self.n_params = sum(int(np.prod(r['shape'])) for r in self.recs)

    def _view(self, flat, r):
        """strided numpy view of one TF variable inside a flat numpy block"""
        shp = r['shape']
        if len(shp) == 1 or r['name'].startswith('dense/'):
            return flat[r['offset']:r['offset'] + int(np.prod(shp))].reshape(shp)
        rows, cols = int(np.prod(shp[:-1])), shp[-1]
        v = np.lib.stride_tricks.as_strided(flat[r['offset']:], shape=(rows, cols),
                                            strides=(r['ld'] * flat.itemsize, flat.itemsize))
        return v  # caller reshapes copies; assignment goes through v[...] = x.reshape(rows, cols)

    def pack(self, tf_vars, scope, dtype=np.float32):
        """dict {'<scope>/<name>': ndarray} -> flat block"""
        flat = np.zeros(self.n_floats, dtype=dtype)
        for r in self.recs:
            a = np.asarray(tf_vars[scope + '/' + r['name']], dtype=dtype)
            assert tuple(a.shape) == r['shape'], (r['name'], a.shape, r['shape'])
            v = self._view(flat, r)
            v[...] = a.reshape(v.shape)
        return flat

    def unpack(self, flat, scope):
        flat = np.ascontiguousarray(flat)
        out = {}
        for r in self.recs:
            out[scope + '/' + r['name']] = np.array(self._view(flat, r)).reshape(r['shape']).copy()
        return out

    def init_block(self, rng, dtype=np.float32):
        """tf.layers defaults (module.py:29,39 pass kernel_initializer=None): glorot-uniform
        kernels with TF fans, zero biases; instance_norm: beta 0, gamma 1."""
        flat = np.zeros(self.n_floats, dtype=dtype)
        for r in self.recs:
            shp, name = r['shape'], r['name']
            v = self._view(flat, r)
            if name.endswith('/kernel'):
                rf = int(np.prod(shp[:-2])) if len(shp) > 2 else 1
                lim = np.sqrt(6.0 / (rf * shp[-2] + rf * shp[-1]))
                v[...] = rng.uniform(-lim, lim, size=v.shape).astype(dtype)
            elif name.endswith('/gamma'):
                v[...] = 1.0
        return flat
