"""TensorFlow on-disk formats the reference produces and consumes, without TensorFlow:

  * checkpoint bundles  -- `tf.train.Saver().save/restore` (reference model.py:21,140-150; convert.py:16):
        <prefix>.index                 an SSTable (LevelDB table format) name -> BundleEntryProto, key "" -> BundleHeaderProto
        <prefix>.data-00000-of-00001   the raw little-endian tensor bytes
        checkpoint                     text file naming the latest prefix
  * event files         -- `tf.summary.FileWriter` (reference model.py:26-29,116,121): TFRecord-framed Event protos
        with Summary.Value{tag, simple_value}, readable by TensorBoard.

Written from the published formats (tensorflow/core/util/tensor_bundle, core/lib/io/{table,format,block}*,
core/lib/io/record_writer, core/framework/summary.proto, core/util/event.proto).  TensorFlow is not installable in the
build container, so these writers/readers are validated against each other, against the formats' known constants
(CRC-32C vectors, the table magic, mask delta) and a hand-assembled fixture -- not against TensorFlow itself.

Host-side utility code: numpy + the library's vc_crc32c.  No device work.
"""
import os
import socket
import struct
import time

import numpy as np

from ._lib import lib

TABLE_MAGIC = 0xdb4775248b80fb57
MASK_DELTA = 0xa282ead8
DT_FLOAT, DT_DOUBLE, DT_INT32, DT_INT64 = 1, 2, 3, 9
_DTYPES = {DT_FLOAT: np.dtype('<f4'), DT_DOUBLE: np.dtype('<f8'), DT_INT32: np.dtype('<i4'), DT_INT64: np.dtype('<i8')}
_DT_OF = {v: k for k, v in _DTYPES.items()}


# ---------------------------------------------------------------------------- checksums
def crc32c(data, crc=0):
    """CRC-32C (Castagnoli) of bytes / a C-contiguous ndarray."""
    if isinstance(data, np.ndarray):
        a = np.ascontiguousarray(data)
        return int(lib.vc_crc32c(a.ctypes.data, a.nbytes, crc)) if a.nbytes else int(crc)
    b = bytes(data)
    return int(lib.vc_crc32c(b, len(b), crc)) if b else int(crc)


def mask_crc(c):
    """the rotation + delta TF/LevelDB apply to a CRC that is stored next to the data it covers"""
    return (((c >> 15) | (c << 17)) + MASK_DELTA) & 0xffffffff


def unmask_crc(m):
    r = (m - MASK_DELTA) & 0xffffffff
    return ((r >> 17) | (r << 15)) & 0xffffffff


# ---------------------------------------------------------------------------- protobuf wire format (the few bits needed)
def _varint(n):
    n &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = n & 0x7f
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _read_varint(buf, pos):
    shift = val = 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7f) << shift
        if not b & 0x80:
            return val, pos
        shift += 7


def _f_varint(field, v):
    return _varint(field << 3) + _varint(v)


def _f_bytes(field, b):
    return _varint((field << 3) | 2) + _varint(len(b)) + bytes(b)


def _f_fixed32(field, v):
    return _varint((field << 3) | 5) + struct.pack('<I', v)


def _f_float(field, v):
    return _varint((field << 3) | 5) + struct.pack('<f', v)


def _f_double(field, v):
    return _varint((field << 3) | 1) + struct.pack('<d', v)


def _parse(buf):
    """-> list of (field, wire_type, value); length-delimited values are returned as bytes"""
    out, pos, buf = [], 0, bytes(buf)
    while pos < len(buf):
        key, pos = _read_varint(buf, pos)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _read_varint(buf, pos)
        elif wt == 1:
            v = buf[pos:pos + 8]; pos += 8
        elif wt == 2:
            n, pos = _read_varint(buf, pos)
            v = buf[pos:pos + n]; pos += n
        elif wt == 5:
            v = buf[pos:pos + 4]; pos += 4
        else:
            raise ValueError('unsupported protobuf wire type %d' % wt)
        out.append((f, wt, v))
    return out


# ---------------------------------------------------------------------------- snappy (read side only; we write uncompressed blocks)
def snappy_decompress(buf):
    buf = bytes(buf)
    n, pos = _read_varint(buf, 0)
    out = bytearray()
    while pos < len(buf):
        tag = buf[pos]; pos += 1
        kind = tag & 3
        if kind == 0:                                    # literal
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], 'little'); pos += nb
            ln += 1
            out += buf[pos:pos + ln]; pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | buf[pos]; pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 2], 'little'); pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], 'little'); pos += 4
        if off == 0 or off > len(out):
            raise ValueError('corrupt snappy stream')
        for _ in range(ln):                              # copies may overlap their own output
            out.append(out[-off])
    if len(out) != n:
        raise ValueError('snappy length mismatch')
    return bytes(out)


# ---------------------------------------------------------------------------- SSTable (LevelDB table format)
class _BlockBuilder:
    def __init__(self, restart_interval=16):
        self.buf, self.restarts, self.n, self.last, self.ri = bytearray(), [0], 0, b'', restart_interval

    def add(self, key, value):
        shared = 0
        if self.n % self.ri == 0 and self.n:
            self.restarts.append(len(self.buf))
        elif self.n:
            m = min(len(key), len(self.last))
            while shared < m and key[shared] == self.last[shared]:
                shared += 1
        self.buf += _varint(shared) + _varint(len(key) - shared) + _varint(len(value)) + key[shared:] + value
        self.last, self.n = key, self.n + 1

    def finish(self):
        return bytes(self.buf) + b''.join(struct.pack('<I', r) for r in self.restarts) + struct.pack('<I', len(self.restarts))


def _block_with_trailer(contents):
    trailer_type = b'\x00'                               # kNoCompression
    c = mask_crc(crc32c(trailer_type, crc32c(contents)))
    return contents + trailer_type + struct.pack('<I', c)


def write_table(path, items, block_size=4096):
    """items: sorted list of (key bytes, value bytes) -> SSTable file (uncompressed blocks)."""
    out = bytearray()
    index = _BlockBuilder(restart_interval=1)
    cur, last_key = _BlockBuilder(), None

    def flush():
        nonlocal cur
        if not cur.n:
            return
        contents = cur.finish()
        handle = _varint(len(out)) + _varint(len(contents))
        out.extend(_block_with_trailer(contents))
        index.add(last_key, handle)                      # any key >= the block's last key and < the next block's first is valid
        cur = _BlockBuilder()

    for k, v in items:
        if last_key is not None and not k > last_key:
            raise ValueError('table keys must be strictly increasing')
        cur.add(k, v)
        last_key = k
        if len(cur.buf) >= block_size:
            flush()
    flush()
    meta = _BlockBuilder().finish()
    meta_handle = _varint(len(out)) + _varint(len(meta))
    out.extend(_block_with_trailer(meta))
    idx = index.finish()
    idx_handle = _varint(len(out)) + _varint(len(idx))
    out.extend(_block_with_trailer(idx))
    footer = meta_handle + idx_handle
    out.extend(footer + b'\x00' * (40 - len(footer)) + struct.pack('<Q', TABLE_MAGIC))
    with open(path, 'wb') as f:
        f.write(out)


def _read_block(buf, offset, size, verify=True):
    contents, typ = buf[offset:offset + size], buf[offset + size]
    if verify:
        want = unmask_crc(struct.unpack('<I', buf[offset + size + 1:offset + size + 5])[0])
        if crc32c(buf[offset + size:offset + size + 1], crc32c(contents)) != want:
            raise ValueError('table block checksum mismatch')
    if typ == 1:
        contents = snappy_decompress(contents)
    elif typ != 0:
        raise ValueError('unknown block compression %d' % typ)
    nrest = struct.unpack('<I', contents[-4:])[0]
    end, pos, key, out = len(contents) - 4 - 4 * nrest, 0, b'', []
    while pos < end:
        shared, pos = _read_varint(contents, pos)
        non_shared, pos = _read_varint(contents, pos)
        vlen, pos = _read_varint(contents, pos)
        key = key[:shared] + contents[pos:pos + non_shared]; pos += non_shared
        out.append((key, contents[pos:pos + vlen])); pos += vlen
    return out


def read_table(path, verify=True):
    buf = open(path, 'rb').read()
    if len(buf) < 48 or struct.unpack('<Q', buf[-8:])[0] != TABLE_MAGIC:
        raise ValueError('%s is not a TensorFlow/LevelDB table (bad magic)' % path)
    footer = buf[-48:]
    _, p = _read_varint(footer, 0); _, p = _read_varint(footer, p)        # metaindex handle (unused)
    ioff, p = _read_varint(footer, p); isz, p = _read_varint(footer, p)
    items = []
    for _, handle in _read_block(buf, ioff, isz, verify):
        off, q = _read_varint(handle, 0); sz, q = _read_varint(handle, q)
        items += _read_block(buf, off, sz, verify)
    return items


# ---------------------------------------------------------------------------- tensor bundle = TF V2 checkpoint
def _shape_proto(shape):
    return b''.join(_f_bytes(2, _f_varint(1, int(d))) for d in shape)        # TensorShapeProto{ repeated Dim dim = 2 { int64 size = 1 } }


def write_tf_checkpoint(prefix, variables):
    """tf.train.Saver().save(sess, prefix) minus the .meta graph: `variables` maps TF variable names to arrays."""
    names = sorted(variables, key=lambda s: s.encode())
    header = _f_varint(1, 1) + _f_varint(2, 0) + _f_bytes(3, _f_varint(1, 1))     # num_shards=1, LITTLE endian, version.producer=1
    items, offset = [(b'', header)], 0
    d = os.path.dirname(prefix)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(prefix + '.data-00000-of-00001', 'wb') as f:
        for n in names:
            a = np.asarray(variables[n])
            dt = a.dtype.newbyteorder('<') if a.dtype.byteorder == '>' else a.dtype
            if np.dtype(dt) not in _DT_OF:
                raise TypeError('%s: dtype %s not supported in a checkpoint' % (n, a.dtype))
            shape = a.shape                                  # (np.ascontiguousarray would turn a 0-d scalar into shape (1,))
            a = np.ascontiguousarray(a, dtype=dt)
            entry = _f_varint(1, _DT_OF[np.dtype(dt)]) + _f_bytes(2, _shape_proto(shape))
            if offset:
                entry += _f_varint(4, offset)
            entry += _f_varint(5, a.nbytes) + _f_fixed32(6, mask_crc(crc32c(a)))   # shard_id 0 is the proto default
            items.append((n.encode(), entry))
            f.write(a.tobytes())
            offset += a.nbytes
    write_table(prefix + '.index', items)
    with open(os.path.join(d, 'checkpoint'), 'w') as f:
        base = os.path.basename(prefix)
        f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (base, base))
    return prefix


def read_tf_checkpoint(prefix, verify=True):
    """-> {variable name: ndarray} from <prefix>.index / <prefix>.data-?????-of-?????"""
    items = read_table(prefix + '.index', verify)
    if not items or items[0][0] != b'':
        raise ValueError('%s.index has no bundle header' % prefix)
    hdr = {f: v for f, _, v in _parse(items[0][1])}
    num_shards = hdr.get(1, 0)
    if hdr.get(2, 0) != 0:
        raise ValueError('big-endian checkpoints are not supported')
    shards, out = {}, {}
    for key, val in items[1:]:
        e = {}
        for f, _, v in _parse(val):
            e.setdefault(f, v)
        if 7 in e:
            raise ValueError('%s: sliced (partitioned) variables are not supported' % key.decode())
        dtype = _DTYPES.get(e.get(1, 0))
        if dtype is None:
            raise TypeError('%s: checkpoint dtype enum %r not supported' % (key.decode(), e.get(1)))
        shape = []
        for f, _, dim in _parse(e.get(2, b'')):
            if f == 2:
                shape.append(next((v for ff, _, v in _parse(dim) if ff == 1), 0))
        shard = e.get(3, 0)
        if shard not in shards:
            fn = '%s.data-%05d-of-%05d' % (prefix, shard, num_shards)
            shards[shard] = np.memmap(fn, dtype=np.uint8, mode='r') if os.path.getsize(fn) else np.zeros(0, np.uint8)   # (an empty file cannot be mapped)
        off, size = e.get(4, 0), e.get(5, 0)
        raw = np.asarray(shards[shard][off:off + size])
        if verify and 6 in e and crc32c(raw) != unmask_crc(struct.unpack('<I', e[6])[0]):
            raise ValueError('%s: tensor checksum mismatch' % key.decode())
        out[key.decode()] = raw.view(dtype).reshape(shape).copy()
    return out


def is_tf_checkpoint(prefix):
    return os.path.exists(prefix + '.index')


# ---------------------------------------------------------------------------- event files (TensorBoard)
class EventFileWriter:
    """tf.summary.FileWriter for scalar summaries: <logdir>/events.out.tfevents.<time>.<host>"""

    def __init__(self, logdir):
        os.makedirs(logdir, exist_ok=True)
        self.path = os.path.join(logdir, 'events.out.tfevents.%010d.%s' % (int(time.time()), socket.gethostname()))
        self._f = open(self.path, 'wb')
        self._record(_f_double(1, time.time()) + _f_bytes(3, b'brain.Event:2'))

    def _record(self, data):
        hdr = struct.pack('<Q', len(data))
        self._f.write(hdr + struct.pack('<I', mask_crc(crc32c(hdr))) + data + struct.pack('<I', mask_crc(crc32c(data))))

    def add_scalars(self, step, scalars, wall_time=None):
        """one Event with a Summary of {tag: value}"""
        summ = b''.join(_f_bytes(1, _f_bytes(1, tag.encode()) + _f_float(2, float(v))) for tag, v in scalars.items())
        self._record(_f_double(1, time.time() if wall_time is None else wall_time) + _f_varint(2, int(step)) + _f_bytes(5, summ))

    def flush(self):
        self._f.flush()

    def close(self):
        self._f.close()


def read_event_file(path, verify=True):
    """-> list of (wall_time, step, {tag: value}) -- the inverse of EventFileWriter, for tests and offline plots"""
    buf, pos, out = open(path, 'rb').read(), 0, []
    while pos < len(buf):
        n = struct.unpack('<Q', buf[pos:pos + 8])[0]
        if verify and unmask_crc(struct.unpack('<I', buf[pos + 8:pos + 12])[0]) != crc32c(buf[pos:pos + 8]):
            raise ValueError('event record length checksum mismatch')
        data = buf[pos + 12:pos + 12 + n]
        if verify and unmask_crc(struct.unpack('<I', buf[pos + 12 + n:pos + 16 + n])[0]) != crc32c(data):
            raise ValueError('event record checksum mismatch')
        pos += 16 + n
        wall, step, vals = 0.0, 0, {}
        for f, _, v in _parse(data):
            if f == 1:
                wall = struct.unpack('<d', v)[0]
            elif f == 2:
                step = v
            elif f == 5:
                for ff, _, val in _parse(v):
                    if ff == 1:
                        d = {a: c for a, _, c in _parse(val)}
                        vals[d[1].decode()] = struct.unpack('<f', d[2])[0]
        out.append((wall, step, vals))
    return out
