"""TensorFlow checkpoint (tensor bundle) reader / writer without TensorFlow.

The reference stores and restores its weights with ``model.save_weights('checkpoints/model')`` /
``model.load_weights('checkpoints/model')`` (reference README.md:66-96, train_sintel.py:104-107,
117-123), i.e. TensorFlow's *tensor bundle*:

    <prefix>.index                   an SSTable in LevelDB table format: tensor name -> BundleEntryProto,
                                     key "" -> BundleHeaderProto
    <prefix>.data-00000-of-00001     the raw little-endian tensor bytes, addressed by (offset, size)

TensorFlow cannot be installed here, so the two formats are restated from their
published layouts (leveldb ``table_format.md``; tensorflow ``core/protobuf/tensor_bundle.proto``,
``core/util/tensor_bundle/tensor_bundle.cc``):

* table file = blocks, each followed by a 5-byte trailer (compression type, masked CRC-32C); footer
  (48 bytes) = metaindex handle, index handle (varint64 offset, size), zero padding to 40 bytes, magic
  ``0xdb4775248b80fb57``; a block = prefix-compressed entries ``varint32 shared | varint32 non_shared |
  varint32 value_len | key delta | value`` + ``uint32 restarts[n] | uint32 n``; compression type 0 = none (what
  TensorFlow's BundleWriter asks for), 1 = snappy (raw block format, decoded by ``snappy_decompress`` -- pinned in the tests
  to the format description's cases and to pyarrow's independent snappy codec);
* ``BundleEntryProto``: 1 dtype, 2 shape (TensorShapeProto: repeated 2 dim {1 size}), 3 shard_id,
  4 offset, 5 size, 6 crc32c (fixed32, masked); ``BundleHeaderProto``: 1 num_shards, 2 endianness, 3 version.

``load_tf_checkpoint(prefix, variant)`` maps the Keras object-graph keys of the reference's ``RAFT`` /
``SmallRAFT`` (``<attribute path>/.ATTRIBUTES/VARIABLE_VALUE``) onto the weight names of
``tf_raft_amd.weights``.  Keras names children of a ``Model`` / ``Sequential`` either by attribute or by
``layer_with_weights-<n>`` / ``layer-<n>`` depending on the version that wrote the file, so every
combination is tried.  ``write_tensor_bundle`` produces files in the same format (name-addressable; no
``_CHECKPOINTABLE_OBJECT_GRAPH``, which Keras' own ``load_weights`` additionally needs): it makes the test
fixtures and lets weights travel back to TensorFlow tooling that reads bundles by name.

**Parity status**: no TensorFlow-written checkpoint is available in this environment (no network, no TF), so
the reader is pinned against the format documents, a bundle assembled byte by byte in the tests, the RFC 3720 CRC-32C
vectors, an independent snappy implementation and its own writer -- not against a file produced by TensorFlow: "parity
unpinned" for real checkpoints.
"""
from __future__ import annotations

import ctypes as C
import itertools
import os
import struct
from collections import OrderedDict
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
_MASK_DELTA = 0xa282ead8
_SUFFIX = '/.ATTRIBUTES/VARIABLE_VALUE'

# tensorflow/core/framework/types.proto
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64,
           10: np.bool_, 17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
_DTYPE_CODES = {np.dtype(v): k for k, v in _DTYPES.items()}
DT_STRING = 7


class CheckpointError(ValueError):
    pass


# ------------------------------------------------------------------------------------------------
# CRC-32C (native helper of the C ABI; pure-Python fallback only if the library cannot be loaded)
# ------------------------------------------------------------------------------------------------
_PY_TABLE: Optional[List[int]] = None


def _crc32c_py(data: bytes, crc: int = 0) -> int:
    global _PY_TABLE
    if _PY_TABLE is None:
        tab = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ (0x82f63b78 if c & 1 else 0)
            tab.append(c)
        _PY_TABLE = tab
    c = crc ^ 0xffffffff
    for b in data:
        c = (c >> 8) ^ _PY_TABLE[(c ^ b) & 0xff]
    return c ^ 0xffffffff


def crc32c(data, crc: int = 0) -> int:
    """CRC-32C of a bytes-like object or a C-contiguous NumPy array."""
    try:
        from . import _ffi
        lib = _ffi.load_library()
    except (RuntimeError, OSError):
        return _crc32c_py(bytes(data), crc)
    if isinstance(data, np.ndarray):
        arr = np.ascontiguousarray(data)
        return int(lib.raft_crc32c(crc, arr.ctypes.data_as(C.c_void_p), arr.nbytes))
    buf = bytes(data)
    return int(lib.raft_crc32c(crc, buf, len(buf)))


def mask_crc(crc: int) -> int:
    """leveldb/TF ``crc32c::Mask``: rotate right by 15 and add a constant."""
    return ((((crc >> 15) | (crc << 17)) & 0xffffffff) + _MASK_DELTA) & 0xffffffff


# ------------------------------------------------------------------------------------------------
# varints and the two protobuf messages
# ------------------------------------------------------------------------------------------------
def _put_varint(n: int) -> bytes:
    out = bytearray()
    while True:
        b = n & 0x7f
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _get_varint(buf: bytes, pos: int) -> Tuple[int, int]:
    shift = result = 0
    while True:
        if pos >= len(buf):
            raise CheckpointError('truncated varint')
        b = buf[pos]
        pos += 1
        result |= (b & 0x7f) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 63:
            raise CheckpointError('varint too long')


def _parse_proto(buf: bytes) -> List[Tuple[int, int, object]]:
    """Minimal protobuf wire parser: list of (field number, wire type, value)."""
    out, pos = [], 0
    while pos < len(buf):
        key, pos = _get_varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _get_varint(buf, pos)
        elif wt == 1:
            val = struct.unpack_from('<Q', buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _get_varint(buf, pos)
            val = buf[pos:pos + n]
            if len(val) != n:
                raise CheckpointError('truncated length-delimited field')
            pos += n
        elif wt == 5:
            val = struct.unpack_from('<I', buf, pos)[0]
            pos += 4
        else:
            raise CheckpointError(f'unsupported protobuf wire type {wt}')
        out.append((field, wt, val))
    return out


def _signed64(v: int) -> int:
    return v - (1 << 64) if v >= 1 << 63 else v


def parse_bundle_entry(buf: bytes) -> Dict[str, object]:
    e = {'dtype': 0, 'shape': [], 'shard_id': 0, 'offset': 0, 'size': 0, 'crc32c': None, 'sliced': False}
    for field, _, val in _parse_proto(buf):
        if field == 1:
            e['dtype'] = val
        elif field == 2:
            for f2, _, v2 in _parse_proto(val):
                if f2 == 2:                                        # TensorShapeProto.Dim
                    size = 0
                    for f3, _, v3 in _parse_proto(v2):
                        if f3 == 1:
                            size = _signed64(v3)
                    e['shape'].append(size)
                elif f2 == 3 and v2:
                    raise CheckpointError('tensor of unknown rank in checkpoint')
        elif field == 3:
            e['shard_id'] = val
        elif field == 4:
            e['offset'] = val
        elif field == 5:
            e['size'] = val
        elif field == 6:
            e['crc32c'] = val
        elif field == 7:
            e['sliced'] = True
    return e


def encode_bundle_entry(dtype_code: int, shape: Iterable[int], offset: int, size: int, crc_masked: int) -> bytes:
    dims = b''.join(b'\x12' + _put_varint(len(d)) + d for d in (b'\x08' + _put_varint(s) for s in shape))
    out = b'\x08' + _put_varint(dtype_code) + b'\x12' + _put_varint(len(dims)) + dims
    if offset:
        out += b'\x20' + _put_varint(offset)                        # shard_id 0 and offset 0 are proto3 defaults
    out += b'\x28' + _put_varint(size) + b'\x35' + struct.pack('<I', crc_masked)
    return out


def encode_bundle_header(num_shards: int = 1) -> bytes:
    version = b'\x08\x01'                                           # VersionDef.producer = 1 (kTensorBundleVersion)
    return b'\x08' + _put_varint(num_shards) + b'\x1a' + _put_varint(len(version)) + version


# ------------------------------------------------------------------------------------------------
# LevelDB table
# ------------------------------------------------------------------------------------------------
def _read_block(buf: bytes, offset: int, size: int, verify: bool) -> bytes:
    if offset + size + 5 > len(buf):
        raise CheckpointError('block handle points past the end of the index file')
    data, ctype = buf[offset:offset + size], buf[offset + size]
    if verify:
        want = struct.unpack_from('<I', buf, offset + size + 1)[0]
        if mask_crc(crc32c(buf[offset:offset + size + 1])) != want:
            raise CheckpointError(f'block at offset {offset}: CRC-32C mismatch')
    if ctype == 1:                    # kSnappyCompression.  TensorFlow's BundleWriter asks for uncompressed index blocks
        return snappy_decompress(data)   # (tensor_bundle.cc: options.compression = kNoCompression); other LevelDB writers do not
    if ctype != 0:
        raise CheckpointError(f'block at offset {offset} has unknown compression type {ctype} (0 = none, 1 = snappy)')
    return data


def snappy_decompress(data: bytes) -> bytes:
    """Raw snappy block format (what LevelDB tables store; format_description.txt of google/snappy): a varint with the
    uncompressed length, then elements tagged by the low two bits of their first byte -- 00 literal (length - 1 in the upper six
    bits, 60..63 = that many minus 59 little-endian length bytes follow), 01 copy with an 11-bit offset (length 4..11), 10 / 11
    copy with a 2- / 4-byte little-endian offset (length 1..64).  Copies may overlap their own output (run-length)."""
    try:
        total, pos = _get_varint(data, 0)
        out = bytearray()
        n = len(data)
        while pos < n:
            tag = data[pos]
            pos += 1
            kind = tag & 3
            if kind == 0:
                ln = tag >> 2
                if ln >= 60:
                    nb = ln - 59
                    if pos + nb > n:
                        raise CheckpointError('snappy: truncated literal length')
                    ln = int.from_bytes(data[pos:pos + nb], 'little')
                    pos += nb
                ln += 1
                if pos + ln > n:
                    raise CheckpointError('snappy: literal runs past the end of the block')
                out += data[pos:pos + ln]
                pos += ln
                continue
            if pos + (1, 2, 4)[kind - 1] > n:
                raise CheckpointError('snappy: truncated copy element')
            if kind == 1:
                ln = 4 + ((tag >> 2) & 7)
                off = ((tag >> 5) << 8) | data[pos]
                pos += 1
            elif kind == 2:
                ln = (tag >> 2) + 1
                off = int.from_bytes(data[pos:pos + 2], 'little')
                pos += 2
            else:
                ln = (tag >> 2) + 1
                off = int.from_bytes(data[pos:pos + 4], 'little')
                pos += 4
            if off == 0 or off > len(out):
                raise CheckpointError('snappy: copy offset outside the data produced so far')
            if off >= ln:
                start = len(out) - off
                out += out[start:start + ln]
            else:                                   # overlapping copy: the pattern of the last `off` bytes repeats
                for _ in range(ln):
                    out.append(out[-off])
        if len(out) != total:
            raise CheckpointError(f'snappy: block decompresses to {len(out)} bytes, its header says {total}')
        return bytes(out)
    except IndexError as exc:
        raise CheckpointError('snappy: truncated block') from exc


def _block_entries(block: bytes) -> List[Tuple[bytes, bytes]]:
    if len(block) < 4:
        raise CheckpointError('block too small')
    n_restarts = struct.unpack_from('<I', block, len(block) - 4)[0]
    limit = len(block) - 4 - 4 * n_restarts
    if limit < 0:
        raise CheckpointError('corrupt restart array')
    out, pos, key = [], 0, b''
    while pos < limit:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        if shared > len(key) or pos + non_shared + vlen > limit:
            raise CheckpointError('corrupt block entry')
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        out.append((key, block[pos:pos + vlen]))
        pos += vlen
    return out


def read_table(path: str, verify: bool = True) -> 'OrderedDict[bytes, bytes]':
    """All key/value pairs of a LevelDB-format table file, in key order."""
    with open(path, 'rb') as f:
        buf = f.read()
    if len(buf) < 48:
        raise CheckpointError(f'{path}: too small to be a table file')
    footer = buf[-48:]
    if struct.unpack('<Q', footer[40:])[0] != TABLE_MAGIC:
        raise CheckpointError(f'{path}: bad table magic (not a TensorFlow checkpoint index)')
    pos = 0
    _, pos = _get_varint(footer, pos)                               # metaindex handle (unused)
    _, pos = _get_varint(footer, pos)
    idx_off, pos = _get_varint(footer, pos)
    idx_size, pos = _get_varint(footer, pos)
    out: 'OrderedDict[bytes, bytes]' = OrderedDict()
    for _, handle in _block_entries(_read_block(buf, idx_off, idx_size, verify)):
        off, p = _get_varint(handle, 0)
        size, _ = _get_varint(handle, p)
        for k, v in _block_entries(_read_block(buf, off, size, verify)):
            out[k] = v
    return out


class _BlockBuilder:
    def __init__(self, restart_interval=16):
        self.buf = bytearray()
        self.restarts = [0]
        self.count = 0
        self.last = b''
        self.interval = restart_interval

    def add(self, key: bytes, value: bytes):
        shared = 0
        if self.count < self.interval:
            n = min(len(key), len(self.last))
            while shared < n and key[shared] == self.last[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.count = 0
        self.buf += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value))
        self.buf += key[shared:] + value
        self.last = key
        self.count += 1

    def finish(self) -> bytes:
        return bytes(self.buf) + b''.join(struct.pack('<I', r) for r in self.restarts) + struct.pack('<I', len(self.restarts))


def write_table(path: str, items: 'Dict[bytes, bytes]', block_size: int = 4096) -> None:
    """Write a LevelDB-format table (uncompressed blocks) with the given key/value pairs."""
    out = bytearray()

    def emit(block: bytes) -> bytes:
        off = len(out)
        out.extend(block)
        out.append(0)                                               # kNoCompression
        out.extend(struct.pack('<I', mask_crc(crc32c(block + b'\x00'))))
        return _put_varint(off) + _put_varint(len(block))

    index = _BlockBuilder(restart_interval=1)
    cur, last_key = _BlockBuilder(), None
    for key in sorted(items):
        cur.add(key, items[key])
        last_key = key
        if len(cur.buf) >= block_size:
            index.add(last_key, emit(cur.finish()))
            cur = _BlockBuilder()
    if cur.buf or last_key is None:
        index.add(last_key if last_key is not None else b'', emit(cur.finish()))
    meta = emit(_BlockBuilder().finish())
    idx = emit(index.finish())
    footer = meta + idx
    out.extend(footer + b'\x00' * (40 - len(footer)) + struct.pack('<Q', TABLE_MAGIC))
    with open(path, 'wb') as f:
        f.write(bytes(out))


# ------------------------------------------------------------------------------------------------
# tensor bundle
# ------------------------------------------------------------------------------------------------
def _data_path(prefix: str, shard: int, num_shards: int) -> str:
    return f'{prefix}.data-{shard:05d}-of-{num_shards:05d}'


def read_tensor_bundle(prefix: str, verify: bool = True) -> 'OrderedDict[str, np.ndarray]':
    """Every numeric tensor of the checkpoint ``prefix`` (string tensors such as
    ``_CHECKPOINTABLE_OBJECT_GRAPH`` are skipped).  ``verify`` checks block and tensor CRC-32Cs."""
    index = prefix + '.index'
    if not os.path.exists(index):
        raise FileNotFoundError(f'{index} not found (expected a TensorFlow checkpoint prefix)')
    table = read_table(index, verify=verify)
    header = table.get(b'')
    if header is None:
        raise CheckpointError(f'{index}: no bundle header entry')
    num_shards, endian = 1, 0
    for field, _, val in _parse_proto(header):
        if field == 1:
            num_shards = val
        elif field == 2:
            endian = val
    if endian != 0:
        raise CheckpointError('big-endian checkpoints are not supported')
    shards: Dict[int, np.memmap] = {}
    out: 'OrderedDict[str, np.ndarray]' = OrderedDict()
    for key, val in table.items():
        if key == b'':
            continue
        e = parse_bundle_entry(val)
        name = key.decode('utf-8')
        if e['dtype'] == DT_STRING:
            continue
        if e['sliced']:
            raise CheckpointError(f'{name}: partitioned (sliced) variables are not supported')
        if e['dtype'] not in _DTYPES:
            raise CheckpointError(f'{name}: unsupported dtype code {e["dtype"]}')
        dt = np.dtype(_DTYPES[e['dtype']])
        count = int(np.prod(e['shape'], dtype=np.int64)) if e['shape'] else 1
        if count * dt.itemsize != e['size']:
            raise CheckpointError(f'{name}: size {e["size"]} does not match shape {e["shape"]} of {dt}')
        sid = e['shard_id']
        if sid not in shards:
            path = _data_path(prefix, sid, num_shards)
            if not os.path.exists(path):
                raise FileNotFoundError(f'{path} not found')
            shards[sid] = np.memmap(path, dtype=np.uint8, mode='r') if os.path.getsize(path) else np.zeros(0, np.uint8)
        raw = shards[sid][e['offset']:e['offset'] + e['size']]
        if raw.size != e['size']:
            raise CheckpointError(f'{name}: data file is truncated')
        raw = np.array(raw)                                         # copy out of the memmap
        if verify and e['crc32c'] is not None and mask_crc(crc32c(raw)) != e['crc32c']:
            raise CheckpointError(f'{name}: tensor CRC-32C mismatch')
        out[name] = raw.view(dt).reshape(e['shape'])
    return out


def write_tensor_bundle(prefix: str, tensors: Dict[str, np.ndarray]) -> None:
    """Write ``tensors`` as ``<prefix>.index`` + ``<prefix>.data-00000-of-00001``."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    items: Dict[bytes, bytes] = {b'': encode_bundle_header(1)}
    offset = 0
    with open(_data_path(prefix, 0, 1), 'wb') as f:
        for name in sorted(tensors):
            arr = np.asarray(tensors[name])
            if arr.ndim and not arr.flags.c_contiguous:
                arr = np.ascontiguousarray(arr)          # (0-d arrays stay 0-d: scalars keep shape ())
            if arr.dtype not in _DTYPE_CODES:
                raise CheckpointError(f'{name}: dtype {arr.dtype} cannot be stored')
            f.write(arr.tobytes())
            items[name.encode('utf-8')] = encode_bundle_entry(_DTYPE_CODES[arr.dtype], arr.shape, offset, arr.nbytes,
                                                              mask_crc(crc32c(arr)))
            offset += arr.nbytes
    write_table(prefix + '.index', items)


# ------------------------------------------------------------------------------------------------
# Keras object-graph keys <-> tf_raft_amd.weights names
# ------------------------------------------------------------------------------------------------
_ROOT_ORDER = ('fnet', 'cnet', 'update_block')                      # attribute order in reference model.py:24-30


def _weighted_index(parts: List[str], i: int, variant: str) -> int:
    """Index among the layers *with weights* of the Sequential that ``parts[i]`` (a layer index) lives in."""
    idx = int(parts[i])
    if parts[i - 1] == 'mask':                                      # Conv2D, ReLU, Conv2D (reference update.py:137-141)
        return {0: 0, 2: 1}[idx]
    return idx                                                       # layerN: two ResBlocks; downsample: conv, norm


def keras_key_candidates(name: str, variant: str = 'raft') -> List[str]:
    """Checkpoint keys (without the ``/.ATTRIBUTES/VARIABLE_VALUE`` suffix) under which Keras may have stored the
    weight ``name`` of ``tf_raft_amd.weights``."""
    parts = name.split('/')
    options: List[List[str]] = []
    for i, p in enumerate(parts):
        if i == 0 and p in _ROOT_ORDER:
            options.append([p, f'layer_with_weights-{_ROOT_ORDER.index(p)}', f'layer-{_ROOT_ORDER.index(p)}'])
        elif p.isdigit():
            k = _weighted_index(parts, i, variant)
            options.append([f'layer_with_weights-{k}', f'layer-{p}', p, f'_layers/{p}', f'layers/{p}'])
        else:
            options.append([p])
    return ['/'.join(c) for c in itertools.product(*options)]


def keras_key(name: str, variant: str = 'raft', style: str = 'attribute') -> str:
    """The key ``write_tf_checkpoint`` stores ``name`` under: ``style='attribute'`` names the root's children
    ``fnet`` / ``cnet`` / ``update_block``, ``style='indexed'`` names them ``layer_with_weights-<n>``; the layers
    of every ``Sequential`` are ``layer_with_weights-<n>`` in both."""
    parts = name.split('/')
    out = []
    for i, p in enumerate(parts):
        if i == 0 and p in _ROOT_ORDER and style == 'indexed':
            out.append(f'layer_with_weights-{_ROOT_ORDER.index(p)}')
        elif p.isdigit():
            out.append(f'layer_with_weights-{_weighted_index(parts, i, variant)}')
        else:
            out.append(p)
    return '/'.join(out)


def map_keras_variables(tensors: Dict[str, np.ndarray], variant: str = 'raft') -> 'OrderedDict[str, np.ndarray]':
    """Pick the model variables out of a checkpoint's tensors and rename them to ``tf_raft_amd.weights`` names."""
    from . import weights as wm
    stripped = {(k[:-len(_SUFFIX)] if k.endswith(_SUFFIX) else k): v for k, v in tensors.items()}
    out: 'OrderedDict[str, np.ndarray]' = OrderedDict()
    missing = []
    for name in wm.init_weights(variant, seed=0):
        for cand in keras_key_candidates(name, variant):
            if cand in stripped:
                out[name] = np.asarray(stripped[cand], dtype=np.float32)
                break
        else:
            missing.append(name)
    if missing:
        model_keys = sorted(k for k in stripped if not k.startswith(('optimizer', 'save_counter', '_CHECKPOINTABLE')))
        raise CheckpointError(f'{len(missing)} weights of {variant!r} not found in the checkpoint, e.g. {missing[:3]}; '
                              f'the checkpoint holds {len(model_keys)} model keys, e.g. {model_keys[:5]}')
    wm.check_weights(variant, out)
    return out


def load_tf_checkpoint(prefix: str, variant: str = 'raft', verify: bool = True) -> 'OrderedDict[str, np.ndarray]':
    """Weights of the reference's ``RAFT`` (``variant='raft'``) or ``SmallRAFT`` (``'small'``) from the TensorFlow
    checkpoint ``prefix`` (e.g. ``'checkpoints/model'``), in Keras layout under ``tf_raft_amd.weights`` names."""
    return map_keras_variables(read_tensor_bundle(prefix, verify=verify), variant)


def write_tf_checkpoint(prefix: str, weights: Dict[str, np.ndarray], variant: str = 'raft', style: str = 'attribute') -> None:
    """Store ``weights`` under Keras object-graph keys in tensor-bundle format (see the module docstring for what
    TensorFlow's own ``load_weights`` would additionally need)."""
    write_tensor_bundle(prefix, {keras_key(k, variant, style) + _SUFFIX: np.asarray(v, np.float32) for k, v in weights.items()})


def is_tf_checkpoint(path: str) -> bool:
    return os.path.exists(path + '.index')
