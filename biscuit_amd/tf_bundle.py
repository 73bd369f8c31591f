"""TensorFlow checkpoint ("tensor bundle", V2) reader and writer without TensorFlow.

A Slideflow / Keras SavedModel directory -- the trained model ``biscuit/utils.py:233-272`` (``find_model``)
hands to ``Project.evaluate(model=...)`` (``biscuit/experiment.py:912-922``) -- keeps its weights in
``variables/variables.index`` + ``variables/variables.data-00000-of-0000N``.  TensorFlow is not vendored by the
reference (``requirements.txt``) and is not installable here, so this module restates the published on-disk
format (tensorflow/core/util/tensor_bundle, tensorflow/core/lib/io/table*, both derived from LevelDB's table
format):

  * the ``.index`` file is an immutable sorted string table: blocks of prefix-compressed
    ``(key, value)`` entries (``varint shared | varint non_shared | varint value_len | key suffix | value``,
    then the restart array and its length as little-endian uint32), each block followed by a one-byte
    compression type (0 = none, 1 = snappy) and the masked CRC-32C of block + type; a 48-byte footer holds
    the block handles (varint offset, varint size) of the meta-index and index blocks and the magic
    0xdb4775248b80fb57;
  * key ``""`` maps to a ``BundleHeaderProto`` (shards, endianness), every other key to a
    ``BundleEntryProto`` (dtype, shape, shard, offset, size, masked CRC-32C of the bytes);
  * tensor bytes sit raw (little endian, row major) in the data shards; string tensors are stored as
    ``varint64 lengths | uint32 masked crc of the lengths | bytes``.

PARITY UNPINNED: there is no TensorFlow-written checkpoint in the container or in the reference to read
back; the tests pin the reader against this module's own writer, against hand-assembled blocks
(prefix compression, restarts, multi-block indexes) and against the CRC / varint known answers of the
format.  Snappy-compressed index blocks (TensorFlow's bundle writer does not produce them) are reported,
not decoded.
"""
import os
import struct

import numpy as np

from .tfrecord import _fields, _varint, crc32c

TABLE_MAGIC = 0xdb4775248b80fb57
FOOTER_LEN = 48
HEADER_KEY = b''
OBJECT_GRAPH_KEY = b'_CHECKPOINTABLE_OBJECT_GRAPH'

# tensorflow/core/framework/types.proto
DTYPES = {1: np.dtype('<f4'), 2: np.dtype('<f8'), 3: np.dtype('<i4'), 4: np.dtype('u1'), 5: np.dtype('<i2'),
          6: np.dtype('i1'), 9: np.dtype('<i8'), 10: np.dtype('?'), 17: np.dtype('<u2'), 19: np.dtype('<f2'),
          22: np.dtype('<u4'), 23: np.dtype('<u8')}
DT_STRING, DT_BFLOAT16 = 7, 14
_DT_OF = {np.dtype(v).newbyteorder('=') if np.dtype(v).byteorder == '<' else np.dtype(v): k for k, v in DTYPES.items()}


class BundleError(ValueError):
    pass


def _guard(fn):
    """A damaged checkpoint must end in BundleError, whatever the parser tripped over (a protobuf wire type that does not exist, a
    varint that runs off the buffer, a shape whose product overflows ...)."""
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **k):
        try:
            return fn(*a, **k)
        except BundleError:
            raise
        except KeyError:
            raise                                                   # (tensor(name) of a name that is not there)
        except (ValueError, IndexError, struct.error, TypeError, OverflowError, MemoryError, UnicodeDecodeError) as e:
            raise BundleError(f'damaged checkpoint: {type(e).__name__}: {e}') from None
    return wrapped


def _mask(crc):
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def _fast_crc():
    """masked CRC-32C of a bytes-like: the native reader's slicing-by-8 when libbiscuit_io.so is built
    (weights are ~90 MB; the pure-Python table walk does ~10 MB/s)."""
    try:
        from . import tfrecord_native
        if tfrecord_native.available():
            fn = tfrecord_native.lib().bqio_masked_crc32c
            return lambda b: int(fn(bytes(b), len(b)))
    except Exception:
        pass
    return lambda b: _mask(crc32c(bytes(b)))


def _put_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


# ---------------------------------------------------------------------------------- sorted string table
def _read_block(buf, offset, size, verify):
    """Entries of the block at [offset, offset+size) (+5-byte trailer) as a list of (key, value) bytes."""
    if offset + size + 5 > len(buf):
        raise BundleError('index file truncated: block runs past the end')
    data = buf[offset:offset + size]
    ctype = buf[offset + size]
    if verify:
        want = struct.unpack_from('<I', buf, offset + size + 1)[0]
        got = _mask(crc32c(bytes(buf[offset:offset + size + 1])))
        if want != got:
            raise BundleError(f'index block at {offset}: CRC mismatch (stored {want:#x}, computed {got:#x})')
    if ctype == 1:
        raise BundleError('snappy-compressed index block (not produced by TensorFlow\'s bundle writer): unsupported')
    if ctype != 0:
        raise BundleError(f'unknown block compression type {ctype}')
    if size < 4:
        raise BundleError('index block too small')
    nrestart = struct.unpack_from('<I', data, size - 4)[0]
    end = size - 4 - 4 * nrestart
    if end < 0:
        raise BundleError('index block: bad restart count')
    out, i, key = [], 0, b''
    while i < end:
        shared, i = _varint(data, i)
        non_shared, i = _varint(data, i)
        vlen, i = _varint(data, i)
        if shared > len(key) or i + non_shared + vlen > end:
            raise BundleError('index block: corrupt entry')
        key = key[:shared] + bytes(data[i:i + non_shared])
        i += non_shared
        out.append((key, bytes(data[i:i + vlen])))
        i += vlen
    return out


def read_table(path, verify=True):
    """{key: value} (bytes -> bytes, in key order) of a TensorFlow/LevelDB-format table file."""
    with open(path, 'rb') as f:
        buf = memoryview(f.read())
    if len(buf) < FOOTER_LEN:
        raise BundleError(f'{path}: too short for a table footer')
    foot = buf[len(buf) - FOOTER_LEN:]
    if struct.unpack_from('<Q', foot, FOOTER_LEN - 8)[0] != TABLE_MAGIC:
        raise BundleError(f'{path}: not a TensorFlow table (bad magic)')
    i = 0
    _, i = _varint(foot, i)            # meta-index handle (unused by the bundle format)
    _, i = _varint(foot, i)
    ioff, i = _varint(foot, i)
    isize, i = _varint(foot, i)
    table = {}
    for _, handle in _read_block(buf, ioff, isize, verify):
        boff, j = _varint(handle, 0)
        bsize, j = _varint(handle, j)
        for k, v in _read_block(buf, boff, bsize, verify):
            table[k] = v
    return table


def _build_block(entries, restart_interval=16):
    out, restarts, prev = bytearray(), [], b''
    for n, (k, v) in enumerate(entries):
        shared = 0
        if n % restart_interval == 0:
            restarts.append(len(out))
        else:
            m = min(len(prev), len(k))
            while shared < m and prev[shared] == k[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
        prev = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack('<I', r)
    out += struct.pack('<I', len(restarts))
    return bytes(out)


def write_table(path, items, block_size=4096, restart_interval=16):
    """Write {key: value} as a table file (uncompressed blocks, like TensorFlow's bundle writer)."""
    entries = sorted(items.items())
    out = bytearray()
    index = []

    def flush(block_entries):
        blk = _build_block(block_entries, restart_interval)
        off = len(out)
        out.extend(blk)
        out.append(0)
        out.extend(struct.pack('<I', _mask(crc32c(blk + b'\x00'))))
        return off, len(blk)

    cur, cur_bytes = [], 0
    for k, v in entries:
        cur.append((k, v))
        cur_bytes += len(k) + len(v) + 3
        if cur_bytes >= block_size:
            off, size = flush(cur)
            index.append((cur[-1][0], _put_varint(off) + _put_varint(size)))
            cur, cur_bytes = [], 0
    if cur or not index:
        off, size = flush(cur)
        index.append((cur[-1][0] if cur else b'', _put_varint(off) + _put_varint(size)))
    moff, msize = flush([])                                   # empty meta-index block
    blk = _build_block(index, 1)                              # index block: every entry is a restart point
    ioff, isize = len(out), len(blk)
    out.extend(blk)
    out.append(0)
    out.extend(struct.pack('<I', _mask(crc32c(blk + b'\x00'))))
    foot = _put_varint(moff) + _put_varint(msize) + _put_varint(ioff) + _put_varint(isize)
    foot += b'\x00' * (FOOTER_LEN - 8 - len(foot))
    out.extend(foot)
    out.extend(struct.pack('<Q', TABLE_MAGIC))
    with open(path, 'wb') as f:
        f.write(out)


# ---------------------------------------------------------------------------------- bundle
def _parse_shape(buf):
    dims = []
    for fn, wt, v in _fields(buf):
        if fn == 2 and wt == 2:
            size = 0
            for f2, w2, v2 in _fields(v):
                if f2 == 1 and w2 == 0:
                    size = v2 - (1 << 64) if v2 >= (1 << 63) else v2
            dims.append(size)
        elif fn == 3 and wt == 0 and v:
            raise BundleError('tensor of unknown rank in a checkpoint')
    return tuple(dims)


def _parse_entry(buf):
    e = {'dtype': 0, 'shape': (), 'shard': 0, 'offset': 0, 'size': 0, 'crc': None, 'sliced': False}
    for fn, wt, v in _fields(memoryview(buf)):
        if fn == 1 and wt == 0:
            e['dtype'] = v
        elif fn == 2 and wt == 2:
            e['shape'] = _parse_shape(v)
        elif fn == 3 and wt == 0:
            e['shard'] = v
        elif fn == 4 and wt == 0:
            e['offset'] = v
        elif fn == 5 and wt == 0:
            e['size'] = v
        elif fn == 6 and wt == 5:
            e['crc'] = struct.unpack('<I', bytes(v))[0]
        elif fn == 7:
            e['sliced'] = True
    return e


def _parse_header(buf):
    h = {'num_shards': 1, 'endianness': 0}
    for fn, wt, v in _fields(memoryview(buf)):
        if fn == 1 and wt == 0:
            h['num_shards'] = v
        elif fn == 2 and wt == 0:
            h['endianness'] = v
    return h


class BundleReader:
    """``BundleReader('model/variables/variables')`` (the prefix; a SavedModel directory or its
    ``variables`` directory is accepted too).  ``keys()`` lists tensor names, ``tensor(name)`` returns a
    numpy array (bfloat16 widened to float32; a string tensor as an object array of bytes)."""

    @_guard
    def __init__(self, prefix, verify=True):
        prefix = resolve_prefix(prefix)
        self.prefix = prefix
        self.verify = verify
        table = read_table(prefix + '.index', verify)
        if HEADER_KEY not in table:
            raise BundleError(f'{prefix}.index: no bundle header entry')
        self.header = _parse_header(table.pop(HEADER_KEY))
        if self.header['endianness'] != 0:
            raise BundleError('big-endian checkpoint: unsupported')
        self.entries = {k.decode('utf-8', 'surrogateescape'): _parse_entry(v) for k, v in table.items()}
        self._shards = {}
        self._crc = None

    def keys(self):
        return list(self.entries)

    def shape(self, name):
        return self.entries[name]['shape']

    def _shard(self, idx):
        if idx not in self._shards:
            path = f"{self.prefix}.data-{idx:05d}-of-{self.header['num_shards']:05d}"
            if not os.path.exists(path):
                raise BundleError(f'missing data shard {path}')
            self._shards[idx] = np.memmap(path, dtype=np.uint8, mode='r')
        return self._shards[idx]

    @_guard
    def raw(self, name):
        e = self.entries.get(name)
        if e is None:
            raise KeyError(name)
        if e['sliced']:
            raise BundleError(f'{name}: partitioned (sliced) variable: unsupported')
        shard = self._shard(e['shard'])
        if e['offset'] + e['size'] > shard.size:
            raise BundleError(f'{name}: data shard truncated')
        raw = shard[e['offset']:e['offset'] + e['size']]
        if self.verify and e['crc'] is not None:
            if self._crc is None:
                self._crc = _fast_crc()
            got = self._crc(raw.tobytes())
            if got != e['crc']:
                raise BundleError(f'{name}: tensor CRC mismatch (stored {e["crc"]:#x}, computed {got:#x})')
        return e, raw

    @_guard
    def tensor(self, name):
        e, raw = self.raw(name)
        n = int(np.prod(e['shape'], dtype=np.int64)) if e['shape'] else 1
        if e['dtype'] == DT_STRING:
            buf = memoryview(raw.tobytes())
            i, lens = 0, []
            for _ in range(n):
                ln, i = _varint(buf, i)
                lens.append(ln)
            if self.verify:
                want = struct.unpack_from('<I', buf, i)[0]
                if _mask(crc32c(bytes(buf[:i]))) != want:
                    raise BundleError(f'{name}: string-length checksum mismatch')
            i += 4
            out = np.empty(n, dtype=object)
            for k, ln in enumerate(lens):
                out[k] = bytes(buf[i:i + ln])
                i += ln
            return out.reshape(e['shape'])
        if e['dtype'] == DT_BFLOAT16:
            bits = np.frombuffer(raw, dtype='<u2', count=n).astype(np.uint32) << 16
            return bits.view(np.float32).reshape(e['shape'])
        dt = DTYPES.get(e['dtype'])
        if dt is None:
            raise BundleError(f'{name}: unsupported dtype enum {e["dtype"]}')
        if n * dt.itemsize != e['size']:
            raise BundleError(f'{name}: {e["size"]} bytes do not match shape {e["shape"]} of {dt}')
        return np.frombuffer(raw, dtype=dt, count=n).reshape(e['shape']).copy()

    def object_graph(self):
        """Serialized ``TrackableObjectGraph`` bytes, or None (name-based checkpoints have none)."""
        if OBJECT_GRAPH_KEY.decode() not in self.entries:
            return None
        return self.tensor(OBJECT_GRAPH_KEY.decode()).reshape(-1)[0]


def resolve_prefix(path):
    """Accept a checkpoint prefix, a ``variables`` directory or a SavedModel directory."""
    if os.path.exists(path + '.index'):
        return path
    for cand in (os.path.join(path, 'variables', 'variables'), os.path.join(path, 'variables')):
        if os.path.exists(cand + '.index'):
            return cand
    if os.path.isdir(path):
        idx = sorted(f for f in os.listdir(path) if f.endswith('.index'))
        if len(idx) == 1:
            return os.path.join(path, idx[0][:-len('.index')])
    raise BundleError(f'{path}: no checkpoint index (looked for <prefix>.index, variables/variables.index)')


def _shape_proto(shape):
    out = b''
    for d in shape:
        dim = b'\x08' + _put_varint(int(d) & ((1 << 64) - 1))
        out += b'\x12' + _put_varint(len(dim)) + dim
    return out


def write_bundle(prefix, tensors, block_size=4096):
    """Write {name: ndarray | bytes} as a one-shard bundle (``bytes`` values become scalar string tensors).
    Used by the tests and by anyone who wants to hand these weights back to TensorFlow."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    items = {HEADER_KEY: b'\x08\x01' + b'\x10\x00' + b'\x1a\x02\x08\x01'}      # 1 shard, little endian, producer 1
    data = bytearray()
    crc = _fast_crc()
    for name in sorted(tensors):
        v = tensors[name]
        if isinstance(v, (bytes, bytearray)):
            lens = _put_varint(len(v))
            raw = lens + struct.pack('<I', _mask(crc32c(lens))) + bytes(v)
            dtype, shape = DT_STRING, ()
        else:
            a = np.asarray(v)
            dtype = _DT_OF.get(a.dtype.newbyteorder('=') if a.dtype.byteorder in '<=' else a.dtype)
            if dtype is None:
                raise BundleError(f'{name}: cannot store dtype {a.dtype}')
            raw, shape = a.astype(a.dtype.newbyteorder('<')).tobytes(order='C'), a.shape
        shp = _shape_proto(shape)
        entry = b'\x08' + _put_varint(dtype) + b'\x12' + _put_varint(len(shp)) + shp
        if len(data):
            entry += b'\x20' + _put_varint(len(data))
        entry += b'\x28' + _put_varint(len(raw)) + b'\x35' + struct.pack('<I', crc(raw))
        items[name.encode()] = entry
        data += raw
    with open(prefix + '.data-00000-of-00001', 'wb') as f:
        f.write(data)
    write_table(prefix + '.index', items, block_size=block_size)


# ---------------------------------------------------------------------------------- object graph
def parse_object_graph(buf):
    """``TrackableObjectGraph`` -> list of nodes ``{'children': {local_name: node_id}, 'attributes':
    {name: checkpoint_key}}`` (node 0 is the root).  tensorflow/core/protobuf/trackable_object_graph.proto."""
    nodes = []
    for fn, wt, v in _fields(memoryview(buf)):
        if fn != 1 or wt != 2:
            continue
        node = {'children': {}, 'attributes': {}}
        for f2, w2, v2 in _fields(v):
            if f2 == 1 and w2 == 2:
                nid, name = 0, ''
                for f3, w3, v3 in _fields(v2):
                    if f3 == 1 and w3 == 0:
                        nid = v3
                    elif f3 == 2 and w3 == 2:
                        name = bytes(v3).decode()
                node['children'][name] = nid
            elif f2 == 2 and w2 == 2:
                name = key = ''
                for f3, w3, v3 in _fields(v2):
                    if f3 == 1 and w3 == 2:
                        name = bytes(v3).decode()
                    elif f3 == 3 and w3 == 2:
                        key = bytes(v3).decode()
                node['attributes'][name] = key
        nodes.append(node)
    return nodes


def _ld(field, payload):
    """One length-delimited protobuf field."""
    return _put_varint(field << 3 | 2) + _put_varint(len(payload)) + bytes(payload)


def build_object_graph(nodes):
    """Inverse of ``parse_object_graph`` (test fixtures, ``keras_import.export_bundle``)."""
    out = b''
    for node in nodes:
        body = b''
        for name, nid in node.get('children', {}).items():
            body += _ld(1, _put_varint(1 << 3) + _put_varint(nid) + _ld(2, name.encode()))
        for name, key in node.get('attributes', {}).items():
            body += _ld(2, _ld(1, name.encode()) + _ld(3, key.encode()))
        out += _ld(1, body)
    return out


# ---------------------------------------------------------------------------------- keras_metadata.pb
def parse_saved_metadata(buf):
    """``keras_metadata.pb`` (tensorflow/python/keras/protobuf/saved_metadata.proto: ``SavedMetadata{repeated SavedObject
    nodes = 1}``, ``SavedObject{node_id = 2, node_path = 3, identifier = 4, metadata = 5 (JSON)}``) ->
    [{'node_id', 'node_path', 'identifier', 'metadata'}]; a metadata string that is not JSON comes back as {}."""
    import json
    out = []
    for fn, wt, v in _fields(memoryview(buf)):
        if fn != 1 or wt != 2:
            continue
        rec = {'node_id': 0, 'node_path': '', 'identifier': '', 'metadata': {}}
        for f2, w2, v2 in _fields(v):
            if f2 == 2 and w2 == 0:
                rec['node_id'] = v2
            elif f2 == 3 and w2 == 2:
                rec['node_path'] = bytes(v2).decode()
            elif f2 == 4 and w2 == 2:
                rec['identifier'] = bytes(v2).decode()
            elif f2 == 5 and w2 == 2:
                try:
                    rec['metadata'] = json.loads(bytes(v2).decode())
                except ValueError:
                    rec['metadata'] = {}
        out.append(rec)
    return out


def build_saved_metadata(records):
    """Inverse of ``parse_saved_metadata`` for [{'node_id', 'node_path', 'identifier', 'metadata': dict}]."""
    import json
    out = b''
    for r in records:
        out += _ld(1, _put_varint(2 << 3) + _put_varint(r['node_id']) + _ld(3, r['node_path'].encode()) +
                   _ld(4, r.get('identifier', '_tf_keras_layer').encode()) + _ld(5, json.dumps(r['metadata']).encode()))
    return out
