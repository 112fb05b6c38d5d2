"""CPU tests of voice_converter_cyclegan_amd.tfio: the TensorFlow on-disk formats the reference reads and writes
(tf.train.Saver checkpoints, model.py:21,140-150; tf.summary.FileWriter event files, model.py:26-29) restated without
TensorFlow.  Pins: CRC-32C vectors of RFC 3720 B.4, the LevelDB/TF table magic, a hand-assembled snappy stream, and
writer<->reader round trips (TensorFlow itself is not installable here -- see the module docstring)."""
import os
import struct

import numpy as np
import pytest

import vcgan_pkg  # noqa: F401


@pytest.fixture(scope='module')
def tfio(built_lib):
    from voice_converter_cyclegan_amd import tfio
    return tfio


def test_crc32c_known_answers(tfio):
    assert tfio.crc32c(b'123456789') == 0xE3069283
    assert tfio.crc32c(bytes(32)) == 0x8A9136AA                      # RFC 3720 B.4
    assert tfio.crc32c(b'\xff' * 32) == 0x62A8AB43
    assert tfio.crc32c(bytes(range(32))) == 0x46DD794E
    assert tfio.crc32c(bytes(range(31, -1, -1))) == 0x113FDB5C
    data = np.random.default_rng(0).integers(0, 256, 100003, dtype=np.uint8)
    whole = tfio.crc32c(data)
    assert tfio.crc32c(data[40001:], tfio.crc32c(data[:40001])) == whole          # streaming == one shot
    assert tfio.crc32c(data.tobytes()) == whole and tfio.crc32c(b'') == 0
    for c in (0, 1, whole, 0xffffffff):
        assert tfio.unmask_crc(tfio.mask_crc(c)) == c
    assert tfio.mask_crc(whole) != whole


def test_snappy_hand_assembled_stream(tfio):
    stream = bytes([12, (4 - 1) << 2]) + b'abcd' + bytes([((8 - 4) << 2) | 1, 4])   # literal 'abcd', copy offset 4 length 8 (overlapping)
    assert tfio.snappy_decompress(stream) == b'abcdabcdabcd'
    long_lit = bytes(range(200)) * 2                                                 # 400-byte literal: 2-byte length form (tag 61)
    stream = tfio._varint(400) + bytes([61 << 2]) + struct.pack('<H', 399) + long_lit
    assert tfio.snappy_decompress(stream) == long_lit
    with pytest.raises(ValueError):
        tfio.snappy_decompress(bytes([5, 0x11, 0x04]))                               # copy before any output


def test_table_round_trip_and_layout(tfio, tmp_path):
    items = [(b'', b'header')] + [(('scope/var_%04d/kernel' % i).encode(), os.urandom(20 + i % 7)) for i in range(700)]
    path = str(tmp_path / 't.index')
    tfio.write_table(path, items, block_size=1024)                    # many data blocks -> a real index block
    buf = open(path, 'rb').read()
    assert struct.unpack('<Q', buf[-8:])[0] == 0xdb4775248b80fb57 and len(buf) > 48
    assert tfio.read_table(path) == items
    # a flipped byte inside a block is caught by the block checksum
    bad = bytearray(buf); bad[10] ^= 1
    open(path, 'wb').write(bad)
    with pytest.raises(ValueError, match='checksum'):
        tfio.read_table(path)
    assert tfio.read_table(path, verify=False) != items
    with pytest.raises(ValueError):
        tfio.write_table(path, [(b'b', b''), (b'a', b'')])            # keys must increase
    # a snappy-compressed block (type byte 1) is accepted on the read side
    blk = tfio._BlockBuilder(); blk.add(b'k1', b'v1'); blk.add(b'k2', b'v2')
    contents = blk.finish()
    comp = tfio._varint(len(contents)) + b''.join(bytes([(min(60, len(contents) - o) - 1) << 2]) + contents[o:o + 60]
                                                   for o in range(0, len(contents), 60))
    body = comp + b'\x01' + struct.pack('<I', tfio.mask_crc(tfio.crc32c(b'\x01', tfio.crc32c(comp))))
    assert tfio._read_block(body, 0, len(comp)) == [(b'k1', b'v1'), (b'k2', b'v2')]


def test_checkpoint_round_trip(tfio, tmp_path):
    rng = np.random.default_rng(1)
    V = {'generator_A2B/h1_conv/kernel': rng.standard_normal((15, 24, 128)).astype(np.float32),
         'generator_A2B/h1_conv/kernel/Adam': rng.standard_normal((15, 24, 128)).astype(np.float32),
         'generator_A2B/h1_conv/bias': rng.standard_normal(128).astype(np.float32),
         'discriminator_A/dense/kernel': rng.standard_normal((1024, 1)).astype(np.float32),
         'beta1_power': np.float32(0.5), 'beta2_power_1': np.float32(0.999),
         'global_step': np.int64(7), 'empty': np.zeros((0, 3), np.float32)}
    prefix = str(tmp_path / 'ckpt' / 'sf1_tf2.ckpt')
    assert tfio.write_tf_checkpoint(prefix, V) == prefix and tfio.is_tf_checkpoint(prefix)
    assert sorted(os.listdir(os.path.dirname(prefix))) == ['checkpoint', 'sf1_tf2.ckpt.data-00000-of-00001', 'sf1_tf2.ckpt.index']
    assert 'model_checkpoint_path: "sf1_tf2.ckpt"' in open(os.path.join(os.path.dirname(prefix), 'checkpoint')).read()
    assert os.path.getsize(prefix + '.data-00000-of-00001') == sum(np.asarray(v).nbytes for v in V.values())
    R = tfio.read_tf_checkpoint(prefix)
    assert sorted(R) == sorted(V)
    for k, v in V.items():
        assert R[k].dtype == np.asarray(v).dtype and R[k].shape == np.asarray(v).shape and np.array_equal(R[k], v), k
    # the bundle header entry: key "", num_shards 1, little endian, version.producer 1
    key, hdr = tfio.read_table(prefix + '.index')[0]
    assert key == b'' and hdr == bytes([0x08, 1, 0x10, 0, 0x1a, 2, 0x08, 1])
    # a corrupted tensor byte is caught by the per-tensor CRC
    with open(prefix + '.data-00000-of-00001', 'r+b') as f:
        f.seek(100); b = f.read(1); f.seek(100); f.write(bytes([b[0] ^ 0x40]))
    with pytest.raises(ValueError, match='checksum'):
        tfio.read_tf_checkpoint(prefix)


def test_event_file_round_trip(tfio, tmp_path):
    w = tfio.EventFileWriter(str(tmp_path))
    w.add_scalars(0, {'generator_summaries/cycle_loss': 1.5, 'generator_summaries/generator_loss': 22.915}, wall_time=1234.5)
    w.add_scalars(1, {'discriminator_summaries/discriminator_loss': 0.049})
    w.close()
    assert os.path.basename(w.path).startswith('events.out.tfevents.')
    ev = tfio.read_event_file(w.path)
    assert len(ev) == 3 and ev[0][2] == {}                            # first record: file_version only
    raw = open(w.path, 'rb').read()
    n0 = struct.unpack('<Q', raw[:8])[0]
    assert b'brain.Event:2' in raw[12:12 + n0]
    assert ev[1][0] == 1234.5 and ev[1][1] == 0 and abs(ev[1][2]['generator_summaries/generator_loss'] - 22.915) < 1e-6
    assert ev[2][1] == 1 and list(ev[2][2]) == ['discriminator_summaries/discriminator_loss']
    bad = bytearray(raw); bad[-6] ^= 1
    open(w.path, 'wb').write(bad)
    with pytest.raises(ValueError, match='checksum'):
        tfio.read_event_file(w.path)


def test_table_bytes_of_a_minimal_table_are_stable(tfio, tmp_path):
    """Byte-level pin of the writer on the smallest table (one key, one block): block entry = varints (shared 0, non-shared 1,
    value length 2) + key + value, one restart at offset 0, restart count 1, trailer = type 0 + masked CRC-32C; then the empty
    metaindex block, the index block, and the 48-byte footer ending in the table magic."""
    path = str(tmp_path / 'one.index')
    tfio.write_table(path, [(b'k', b'vv')])
    buf = open(path, 'rb').read()
    block = bytes([0, 1, 2]) + b'k' + b'vv' + struct.pack('<II', 0, 1)
    assert buf[:len(block)] == block and buf[len(block)] == 0
    assert struct.unpack('<I', buf[len(block) + 1:len(block) + 5])[0] == tfio.mask_crc(tfio.crc32c(b'\x00', tfio.crc32c(block)))
    meta_off = len(block) + 5
    assert buf[meta_off:meta_off + 8] == struct.pack('<II', 0, 1)                     # empty metaindex block: restart array only
    footer = buf[-48:]
    assert footer[0] == meta_off and footer[1] == 8                                   # metaindex handle (offset, size) as varints
    assert footer[2] == meta_off + 8 + 5                                              # index block follows the metaindex block + trailer
    assert footer[40:] == struct.pack('<Q', 0xdb4775248b80fb57) and len(buf) == footer[2] + footer[3] + 5 + 48


def test_checkpoint_round_trip_random(tfio, tmp_path):
    """hypothesis: arbitrary variable names / shapes / dtypes survive write -> read, in key order, with scalars and empties."""
    from hypothesis import given, settings, strategies as st
    names = st.text(alphabet='abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ_/0123456789', min_size=1, max_size=40)
    shapes = st.lists(st.integers(0, 5), min_size=0, max_size=4)
    dts = st.sampled_from([np.float32, np.float64, np.int32, np.int64])
    counter = [0]

    @settings(max_examples=25, deadline=None)
    @given(st.dictionaries(names, st.tuples(shapes, dts), min_size=1, max_size=40), st.integers(0, 2 ** 31 - 1))
    def run(spec, seed):
        rng = np.random.default_rng(seed)
        V = {n: (rng.standard_normal(sh) * 100).astype(dt) for n, (sh, dt) in spec.items()}
        counter[0] += 1
        prefix = str(tmp_path / ('c%d' % counter[0]) / 'm.ckpt')
        tfio.write_tf_checkpoint(prefix, V)
        R = tfio.read_tf_checkpoint(prefix)
        assert list(R) == sorted(V, key=lambda s: s.encode())
        for k, v in V.items():
            assert R[k].dtype == v.dtype and R[k].shape == v.shape and np.array_equal(R[k], v)

    run()
