"""TensorFlow checkpoint (V2 "tensor bundle") reader / writer, pure numpy -- no TensorFlow needed.

The reference keeps model state as ``checkpoints/<name>/model-<step>.{index,data-00000-of-00001,meta}``
plus a ``checkpoint`` state file (``tf.train.Saver(max_to_keep=5)``, reference lib/models.py:351; saved
:924, restored :209-215 through ``tf.train.latest_checkpoint``).  The pretrained models the reference's
README points to are such bundles, with the variable names of SURVEY Appendix B.  This module reads them
into ``{name: ndarray}`` for ``CAPE.load_variables`` and writes the same format so that a model trained
here can be handed back to the TF code.

Format (restated from the published TensorFlow sources, tensorflow/core/util/tensor_bundle and
tensorflow/core/lib/io/{table,block,format}; r1.13 is the reference's pinned version):

* ``<prefix>.index`` is an immutable sorted string table in the LevelDB table layout: data blocks of
  prefix-compressed entries ``varint32 shared | varint32 non_shared | varint32 value_len | key delta |
  value`` followed by a ``uint32`` restart array and its length; every block carries a 5-byte trailer
  (compression type, masked CRC-32C of contents+type); an index block maps separator keys to
  ``BlockHandle(offset, size)``; the 48-byte footer holds the metaindex and index handles and the magic
  ``0xdb4775248b80fb57``.  TensorFlow writes the table uncompressed; Snappy blocks are decoded too.
* key ``""`` -> ``BundleHeaderProto`` (num_shards, endianness, version); every other key is a tensor name
  -> ``BundleEntryProto`` (dtype, shape, shard_id, offset, size, masked crc32c of the bytes).
* ``<prefix>.data-0000i-of-0000N`` hold the raw little-endian tensor bytes at ``offset``.

No TensorFlow is installed in this environment, so the reader is validated against this module's own
writer, against the RFC 3720 CRC-32C vectors and against hand-assembled table bytes (tests/): the byte
layout has not been checked against a file produced by TensorFlow itself.
"""
import os
import re
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
_MASK_DELTA = 0xa282ead8
_BLOCK_SIZE = 262144                   # table::Options::block_size
_RESTART_INTERVAL = 16                 # table::Options::block_restart_interval

# tensorflow/core/framework/types.proto
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64,
           10: np.bool_, 17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
_DTYPE_IDS = {np.dtype(v): k for k, v in _DTYPES.items()}
DT_STRING, DT_BFLOAT16 = 7, 14


class CheckpointError(ValueError):
    pass


# --------------------------------------------------------------------------------------------------
# CRC-32C (Castagnoli, reflected polynomial 0x82F63B78), masked as in tensorflow/core/lib/hash/crc32c.h
# --------------------------------------------------------------------------------------------------
def _crc_table():
    t = np.arange(256, dtype=np.uint32)
    for _ in range(8):
        t = np.where(t & 1, (t >> 1) ^ np.uint32(0x82F63B78), t >> 1).astype(np.uint32)
    return t


_T = _crc_table()
_T_LIST = [int(v) for v in _T]


def _crc_small(buf, state):
    for b in bytes(buf):
        state = _T_LIST[(state ^ b) & 0xFF] ^ (state >> 8)
    return state


def crc32c(data):
    """CRC-32C of a bytes-like object or array.  Large inputs are cut into equal chunks whose registers
    advance together (one numpy step per byte column); the chunk registers are then chained with the
    'append n zero bytes' operator, which is linear over GF(2) and tabulated in the same sweep."""
    if isinstance(data, np.ndarray):
        b = np.ascontiguousarray(data).reshape(-1).view(np.uint8)
    else:
        b = np.frombuffer(data, dtype=np.uint8)
    n = b.size
    if n < 2048:
        return _crc_small(b.tobytes(), 0xFFFFFFFF) ^ 0xFFFFFFFF
    P = int(min(4096, n // 256))
    L = -(-n // P)
    pad = P * L - n
    buf = np.zeros(P * L, dtype=np.uint8)
    buf[pad:] = b
    buf[pad:pad + 4] ^= 0xFF            # initial register folded into the first four message bytes;
    #                                     the zero padding in front of them leaves a zero register at zero
    cols = np.ascontiguousarray(buf.reshape(P, L).T)
    st = np.zeros(P + 1024, dtype=np.uint32)
    st[P:] = (np.arange(256, dtype=np.uint32)[None, :] << (8 * np.arange(4, dtype=np.uint32))[:, None]).reshape(-1)
    for j in range(L):
        st[:P] ^= cols[j]
        st = _T[st & 0xFF] ^ (st >> 8)
    z = [[int(v) for v in st[P + 256 * q:P + 256 * (q + 1)]] for q in range(4)]
    s = 0
    for r in st[:P].tolist():
        s = z[0][s & 0xFF] ^ z[1][(s >> 8) & 0xFF] ^ z[2][(s >> 16) & 0xFF] ^ z[3][s >> 24] ^ r
    return s ^ 0xFFFFFFFF


def crc_mask(crc):
    return (((crc >> 15) | (crc << 17)) + _MASK_DELTA) & 0xFFFFFFFF


def crc_unmask(masked):
    rot = (masked - _MASK_DELTA) & 0xFFFFFFFF
    return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


# --------------------------------------------------------------------------------------------------
# varints and the three small protos
# --------------------------------------------------------------------------------------------------
def _put_varint(v):
    if v < 0:
        v += 1 << 64
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _get_varint(buf, pos):
    shift = result = 0
    while True:
        if pos >= len(buf):
            raise CheckpointError("truncated varint")
        c = buf[pos]
        pos += 1
        result |= (c & 0x7F) << shift
        if not c & 0x80:
            return result, pos
        shift += 7
        if shift > 63:
            raise CheckpointError("varint too long")


def _proto_fields(buf):
    """[(field number, wire type, value)] of one serialized message (value: int or bytes)."""
    pos, out = 0, []
    while pos < len(buf):
        tag, pos = _get_varint(buf, pos)
        fno, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v, pos = struct.unpack_from('<Q', buf, pos)[0], pos + 8
        elif wt == 2:
            ln, pos = _get_varint(buf, pos)
            if pos + ln > len(buf):
                raise CheckpointError("truncated proto field")
            v, pos = bytes(buf[pos:pos + ln]), pos + ln
        elif wt == 5:
            v, pos = struct.unpack_from('<I', buf, pos)[0], pos + 4
        else:
            raise CheckpointError("unsupported proto wire type %d" % wt)
        out.append((fno, wt, v))
    return out


def _field(fno, wt, payload):
    return _put_varint((fno << 3) | wt) + payload


def _signed64(v):
    return v - (1 << 64) if v >= 1 << 63 else v


def _encode_shape(shape):
    out = b''
    for s in shape:
        dim = _field(1, 0, _put_varint(int(s))) if s else b''          # TensorShapeProto.Dim.size
        out += _field(2, 2, _put_varint(len(dim)) + dim)                # TensorShapeProto.dim
    return out


def _decode_shape(buf):
    dims = []
    for fno, wt, v in _proto_fields(buf):
        if fno == 2 and wt == 2:
            size = 0
            for f2, w2, v2 in _proto_fields(v):
                if f2 == 1 and w2 == 0:
                    size = _signed64(v2)
            dims.append(size)
        elif fno == 3 and v:
            raise CheckpointError("tensor of unknown rank in checkpoint")
    return tuple(dims)


def encode_entry(dtype_id, shape, shard_id, offset, size, crc_masked):
    sh = _encode_shape(shape)
    out = _field(1, 0, _put_varint(dtype_id)) + _field(2, 2, _put_varint(len(sh)) + sh)
    if shard_id:
        out += _field(3, 0, _put_varint(shard_id))
    if offset:
        out += _field(4, 0, _put_varint(offset))
    if size:
        out += _field(5, 0, _put_varint(size))
    if crc_masked:
        out += _field(6, 5, struct.pack('<I', crc_masked))
    return out


def decode_entry(buf):
    e = {'dtype': 0, 'shape': (), 'shard_id': 0, 'offset': 0, 'size': 0, 'crc32c': 0, 'slices': 0}
    for fno, wt, v in _proto_fields(buf):
        if fno == 1:
            e['dtype'] = v
        elif fno == 2:
            e['shape'] = _decode_shape(v)
        elif fno == 3:
            e['shard_id'] = v
        elif fno == 4:
            e['offset'] = _signed64(v)
        elif fno == 5:
            e['size'] = _signed64(v)
        elif fno == 6:
            e['crc32c'] = v
        elif fno == 7:
            e['slices'] += 1
    return e


def encode_header(num_shards=1):
    return _field(1, 0, _put_varint(num_shards)) + _field(3, 2, b'\x02' + _field(1, 0, _put_varint(1)))


def decode_header(buf):
    h = {'num_shards': 0, 'endianness': 0, 'producer': 0, 'min_consumer': 0}
    for fno, wt, v in _proto_fields(buf):
        if fno == 1:
            h['num_shards'] = v
        elif fno == 2:
            h['endianness'] = v
        elif fno == 3:
            for f2, w2, v2 in _proto_fields(v):
                if f2 == 1:
                    h['producer'] = v2
                elif f2 == 2:
                    h['min_consumer'] = v2
    return h


# --------------------------------------------------------------------------------------------------
# Snappy block decoder (raw format; only met if some other tool wrote the index compressed)
# --------------------------------------------------------------------------------------------------
def snappy_uncompress(src):
    n, pos = _get_varint(src, 0)
    out = bytearray()
    while pos < len(src):
        tag = src[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(src[pos:pos + nb], 'little')
                pos += nb
            ln += 1
            out += src[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | src[pos]
            pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = src[pos] | (src[pos + 1] << 8)
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(src[pos:pos + 4], 'little')
            pos += 4
        if off == 0 or off > len(out):
            raise CheckpointError("corrupt snappy block")
        for _ in range(ln):                 # copies may overlap their own output
            out.append(out[-off])
    if len(out) != n:
        raise CheckpointError("snappy length mismatch")
    return bytes(out)


# --------------------------------------------------------------------------------------------------
# table blocks
# --------------------------------------------------------------------------------------------------
class _BlockBuilder(object):
    def __init__(self, restart_interval=_RESTART_INTERVAL):
        self.buf, self.restarts, self.counter, self.last_key = bytearray(), [0], 0, b''
        self.interval = restart_interval

    def add(self, key, value):
        shared = 0
        if self.counter < self.interval:
            m = min(len(key), len(self.last_key))
            while shared < m and key[shared] == self.last_key[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.counter = 0
        self.buf += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value))
        self.buf += key[shared:] + value
        self.last_key = key
        self.counter += 1

    def size_estimate(self):
        return len(self.buf) + 4 * len(self.restarts) + 4

    def empty(self):
        return not self.buf

    def finish(self):
        return bytes(self.buf) + b''.join(struct.pack('<I', r) for r in self.restarts) + \
            struct.pack('<I', len(self.restarts))


def _block_entries(contents):
    if len(contents) < 4:
        raise CheckpointError("table block too short")
    nrestarts = struct.unpack_from('<I', contents, len(contents) - 4)[0]
    limit = len(contents) - 4 - 4 * nrestarts
    if limit < 0:
        raise CheckpointError("bad restart array")
    pos, key, out = 0, b'', []
    while pos < limit:
        shared, pos = _get_varint(contents, pos)
        non_shared, pos = _get_varint(contents, pos)
        vlen, pos = _get_varint(contents, pos)
        if shared > len(key) or pos + non_shared + vlen > limit:
            raise CheckpointError("corrupt table entry")
        key = key[:shared] + bytes(contents[pos:pos + non_shared])
        pos += non_shared
        out.append((key, bytes(contents[pos:pos + vlen])))
        pos += vlen
    return out


def _handle(offset, size):
    return _put_varint(offset) + _put_varint(size)


def write_table(path, items, block_size=_BLOCK_SIZE):
    """items: iterable of (key bytes, value bytes) in strictly increasing key order."""
    with open(path, 'wb') as f:
        offset = 0

        def emit(contents):
            nonlocal offset
            trailer = b'\x00'
            f.write(contents + trailer + struct.pack('<I', crc_mask(crc32c(contents + trailer))))
            h = _handle(offset, len(contents))
            offset += len(contents) + 5
            return h

        data, index, last = _BlockBuilder(), _BlockBuilder(1), None
        for key, value in items:
            if last is not None and not key > last:
                raise CheckpointError("table keys must be strictly increasing")
            data.add(key, value)
            last = key
            if data.size_estimate() >= block_size:
                index.add(last, emit(data.finish()))
                data = _BlockBuilder()
        if not data.empty():
            index.add(last, emit(data.finish()))
        meta_h = emit(_BlockBuilder().finish())
        index_h = emit(index.finish())
        footer = meta_h + index_h
        f.write(footer + b'\x00' * (40 - len(footer)) + struct.pack('<Q', TABLE_MAGIC))


def read_table(path, verify=True):
    """[(key, value)] of a table file, in key order."""
    with open(path, 'rb') as f:
        raw = f.read()
    if len(raw) < 48 or struct.unpack_from('<Q', raw, len(raw) - 8)[0] != TABLE_MAGIC:
        raise CheckpointError("%s is not a TensorFlow checkpoint index (bad table magic)" % path)
    footer = raw[-48:]
    _, p = _get_varint(footer, 0)
    _, p = _get_varint(footer, p)
    ioff, p = _get_varint(footer, p)
    isize, p = _get_varint(footer, p)

    def block(off, size):
        if off + size + 5 > len(raw):
            raise CheckpointError("table block out of range")
        contents, kind = raw[off:off + size], raw[off + size]
        if verify:
            want = crc_unmask(struct.unpack_from('<I', raw, off + size + 1)[0])
            if crc32c(raw[off:off + size + 1]) != want:
                raise CheckpointError("table block checksum mismatch at offset %d" % off)
        if kind == 1:
            contents = snappy_uncompress(contents)
        elif kind != 0:
            raise CheckpointError("unknown block compression %d" % kind)
        return contents

    out = []
    for _, hv in _block_entries(block(ioff, isize)):
        off, p = _get_varint(hv, 0)
        size, p = _get_varint(hv, p)
        out.extend(_block_entries(block(off, size)))
    return out


# --------------------------------------------------------------------------------------------------
# bundle level
# --------------------------------------------------------------------------------------------------
def _data_path(prefix, shard, num_shards):
    return "%s.data-%05d-of-%05d" % (prefix, shard, num_shards)


class BundleReader(object):
    """``tf.train.load_checkpoint`` without TensorFlow: ``keys()``, ``shape(name)``, ``get_tensor(name)``."""

    def __init__(self, prefix, verify=True):
        self.prefix, self.verify = prefix, verify
        items = read_table(prefix + '.index', verify)
        if not items or items[0][0] != b'':
            raise CheckpointError("%s.index has no bundle header entry" % prefix)
        self.header = decode_header(items[0][1])
        if self.header['endianness'] != 0:
            raise CheckpointError("big-endian bundles are not supported")
        if self.header['min_consumer'] > 1:
            raise CheckpointError("bundle needs a newer reader (min_consumer %d)" % self.header['min_consumer'])
        self.entries = {k.decode('utf-8'): decode_entry(v) for k, v in items[1:]}

    def keys(self):
        return sorted(self.entries)

    def has_tensor(self, name):
        return name in self.entries

    def shape(self, name):
        return self.entries[name]['shape']

    def variable_to_shape_map(self):
        return {k: list(e['shape']) for k, e in self.entries.items()}

    def get_tensor(self, name):
        if name not in self.entries:
            raise KeyError("tensor %r not found in checkpoint %s" % (name, self.prefix))
        e = self.entries[name]
        if e['slices']:
            raise CheckpointError("%s is a partitioned variable (slices are not supported)" % name)
        if e['dtype'] == DT_STRING:
            raise CheckpointError("%s: string tensors are not supported" % name)
        if e['dtype'] == DT_BFLOAT16:
            dt, widen = np.dtype(np.uint16), True
        elif e['dtype'] in _DTYPES:
            dt, widen = np.dtype(_DTYPES[e['dtype']]), False
        else:
            raise CheckpointError("%s: unsupported dtype enum %d" % (name, e['dtype']))
        count = int(np.prod(e['shape'], dtype=np.int64)) if e['shape'] else 1
        if count * dt.itemsize != e['size']:
            raise CheckpointError("%s: %d bytes stored, shape %s needs %d" % (name, e['size'], e['shape'],
                                                                              count * dt.itemsize))
        path = _data_path(self.prefix, e['shard_id'], max(self.header['num_shards'], 1))
        with open(path, 'rb') as f:
            f.seek(e['offset'])
            raw = f.read(e['size'])
        if len(raw) != e['size']:
            raise CheckpointError("%s: data file %s is truncated" % (name, path))
        if self.verify and crc32c(raw) != crc_unmask(e['crc32c']):
            raise CheckpointError("%s: tensor checksum mismatch" % name)
        arr = np.frombuffer(raw, dtype=dt.newbyteorder('<')).reshape(e['shape'])
        if widen:                                   # bfloat16 -> float32
            arr = (arr.astype(np.uint32) << 16).view(np.float32)
        return np.array(arr)

    def readable(self, name):
        """True if ``get_tensor(name)`` can decode the entry: a whole (unsliced) tensor of a numeric dtype.  String
        tensors (e.g. the ``_CHECKPOINTABLE_OBJECT_GRAPH`` entry newer TF 1.x savers add) and partitioned variables
        are not."""
        e = self.entries.get(name)
        return (e is not None and not e['slices'] and e['dtype'] != DT_STRING and
                (e['dtype'] == DT_BFLOAT16 or e['dtype'] in _DTYPES))

    def read_all(self, names=None, skip_unreadable=True):
        """``{name: array}`` of the checkpoint.  ``names``: optional filter (iterable or predicate).  Entries that
        cannot be decoded (string tensors, slices, unknown dtypes) are skipped -- a checkpoint that carries one must
        still yield its float variables -- unless ``skip_unreadable`` is False, which raises like ``get_tensor``."""
        if names is None:
            want = lambda k: True
        elif callable(names):
            want = names
        else:
            wanted = set(names)
            want = wanted.__contains__
        out = {}
        for k in self.keys():
            if not want(k):
                continue
            if skip_unreadable and not self.readable(k):
                continue
            out[k] = self.get_tensor(k)
        return out


def write_bundle(prefix, arrays):
    """Write ``{name: ndarray}`` as a single-shard bundle (names sorted, tensors packed back to back, as
    ``BundleWriter`` does)."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    items, offset = [(b'', encode_header(1))], 0
    with open(_data_path(prefix, 0, 1), 'wb') as f:
        for name in sorted(arrays, key=lambda s: s.encode('utf-8')):
            if not name:
                raise CheckpointError("empty tensor name")
            a = np.asarray(arrays[name])
            if a.dtype not in _DTYPE_IDS:
                raise CheckpointError("%s: dtype %s cannot be stored" % (name, a.dtype))
            raw = np.ascontiguousarray(a.astype(a.dtype.newbyteorder('<'), copy=False)).tobytes()
            f.write(raw)
            items.append((name.encode('utf-8'),
                          encode_entry(_DTYPE_IDS[a.dtype], a.shape, 0, offset, len(raw), crc_mask(crc32c(raw)))))
            offset += len(raw)
    write_table(prefix + '.index', items)
    return prefix


# --------------------------------------------------------------------------------------------------
# the `checkpoint` state file (CheckpointState text proto)
# --------------------------------------------------------------------------------------------------
def latest_checkpoint(checkpoint_dir):
    """``tf.train.latest_checkpoint``: the prefix named by ``model_checkpoint_path`` if its index exists."""
    state = os.path.join(checkpoint_dir, 'checkpoint')
    if not os.path.isfile(state):
        return None
    m = re.search(r'^\s*model_checkpoint_path:\s*"((?:[^"\\]|\\.)*)"', open(state).read(), flags=re.M)
    if not m:
        return None
    path = m.group(1).replace('\\"', '"').replace('\\\\', '\\')
    if not os.path.isabs(path):
        path = os.path.join(checkpoint_dir, path)
    return path if os.path.isfile(path + '.index') else None


def update_checkpoint_state(checkpoint_dir, prefix, keep=5):
    """Record ``prefix`` as the latest checkpoint (relative paths, like the Saver) and drop the files of
    all but the newest ``keep`` (``max_to_keep``)."""
    state = os.path.join(checkpoint_dir, 'checkpoint')
    known = []
    if os.path.isfile(state):
        known = re.findall(r'^\s*all_model_checkpoint_paths:\s*"([^"]*)"', open(state).read(), flags=re.M)
    name = os.path.relpath(prefix, checkpoint_dir)
    known = [k for k in known if k != name] + [name]
    for old in known[:-keep]:
        base = old if os.path.isabs(old) else os.path.join(checkpoint_dir, old)
        for fn in [base + '.index', base + '.meta'] + \
                [os.path.join(os.path.dirname(base), f) for f in os.listdir(os.path.dirname(base) or '.')
                 if f.startswith(os.path.basename(base) + '.data-')]:
            if os.path.isfile(fn):
                os.remove(fn)
    known = known[-keep:]
    with open(state, 'w') as f:
        f.write('model_checkpoint_path: "%s"\n' % name)
        for k in known:
            f.write('all_model_checkpoint_paths: "%s"\n' % k)


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="list the tensors of a TensorFlow V2 checkpoint")
    ap.add_argument('prefix', help="checkpoint prefix (e.g. checkpoints/<name>/model-1234) or its directory")
    ap.add_argument('--no-verify', action='store_true')
    a = ap.parse_args(argv)
    prefix = a.prefix
    if os.path.isdir(prefix):
        prefix = latest_checkpoint(prefix)
        if prefix is None:
            raise SystemExit("no checkpoint state in %s" % a.prefix)
    r = BundleReader(prefix, verify=not a.no_verify)
    total = 0
    for k in r.keys():
        e = r.entries[k]
        total += e['size']
        print("%-80s %-10s %s" % (k, np.dtype(_DTYPES.get(e['dtype'], np.void)).name, list(e['shape'])))
    print("# %d tensors, %.2f MB, %d shard(s)" % (len(r.entries), total / 1e6, r.header['num_shards']))


if __name__ == '__main__':
    main()
