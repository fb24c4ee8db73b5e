"""TensorFlow V2 checkpoint reader/writer (cape_amd/tf_checkpoint.py): CRC-32C against the RFC 3720 vectors,
the table reader against bytes assembled by hand from the LevelDB table layout, bundle round trips, the
`checkpoint` state file, and corruption handling.  CPU only."""
import os
import struct

import numpy as np
import pytest

from cape_amd import tf_checkpoint as tc


def test_crc32c_known_answers():
    assert tc.crc32c(b"") == 0
    assert tc.crc32c(b"123456789") == 0xE3069283
    assert tc.crc32c(bytes(32)) == 0x8A9136AA                         # RFC 3720 B.4
    assert tc.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert tc.crc32c(bytes(range(32))) == 0x46DD794E
    assert tc.crc32c(bytes(range(31, -1, -1))) == 0x113FDB5C
    # TensorFlow's crc32c_test: masking is not the identity and round-trips
    c = tc.crc32c(b"foo")
    assert tc.crc_mask(c) != c and tc.crc_mask(tc.crc_mask(c)) != c
    assert tc.crc_unmask(tc.crc_mask(c)) == c and tc.crc_unmask(tc.crc_unmask(tc.crc_mask(tc.crc_mask(c)))) == c


def test_crc32c_chunked_sweep_equals_bytewise():
    rng = np.random.default_rng(0)
    for n in (2047, 2048, 2049, 4097, 70001, 1 << 20, (1 << 20) + 3):
        d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert tc.crc32c(d) == tc._crc_small(d, 0xFFFFFFFF) ^ 0xFFFFFFFF, n
    a = rng.standard_normal((300, 7)).astype(np.float32)
    assert tc.crc32c(a) == tc.crc32c(a.tobytes()) == tc.crc32c(np.asfortranarray(a))


def _hand_table(entries_by_block):
    """LevelDB table bytes written out longhand (no shared code with the writer under test): entries without
    prefix sharing, one restart per block, uncompressed blocks."""
    def varint(v):
        out = b''
        while v >= 0x80:
            out += bytes([(v & 0x7F) | 0x80])
            v >>= 7
        return out + bytes([v])

    def block(entries):
        body = b''
        for k, v in entries:
            body += varint(0) + varint(len(k)) + varint(len(v)) + k + v
        body += struct.pack('<I', 0) + struct.pack('<I', 1)
        crc = tc._crc_small(body + b'\x00', 0xFFFFFFFF) ^ 0xFFFFFFFF
        masked = (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xFFFFFFFF
        return body, body + b'\x00' + struct.pack('<I', masked)

    out, index = b'', []
    for entries in entries_by_block:
        body, raw = block(entries)
        index.append((entries[-1][0], varint(len(out)) + varint(len(body))))
        out += raw
    body, raw = block([])
    meta = varint(len(out)) + varint(len(body))
    out += raw
    body, raw = block(index)
    idx = varint(len(out)) + varint(len(body))
    out += raw
    footer = meta + idx
    return out + footer + b'\x00' * (40 - len(footer)) + bytes.fromhex('57fb808b247547db')


def test_table_reader_on_hand_assembled_bytes(tmp_path):
    blocks = [[(b'', b'hdr'), (b'a/b', b'1'), (b'a/bc', b'22')], [(b'b', b''), (b'c' * 300, b'x' * 1000)]]
    fn = str(tmp_path / 't.index')
    open(fn, 'wb').write(_hand_table(blocks))
    assert tc.read_table(fn) == [kv for b in blocks for kv in b]
    raw = bytearray(open(fn, 'rb').read())
    raw[3] ^= 1                                            # flip one bit inside the first data block
    open(fn, 'wb').write(bytes(raw))
    with pytest.raises(tc.CheckpointError, match="checksum"):
        tc.read_table(fn)
    open(fn, 'wb').write(bytes(raw[:-1]) + b'\x00')        # break the magic
    with pytest.raises(tc.CheckpointError, match="magic"):
        tc.read_table(fn)


def test_table_writer_prefix_compression_restarts_and_blocks(tmp_path):
    keys = sorted(("generator/decoder/decoder_resblock_affine%d/%s/weights" % (i, n)).encode()
                  for i in range(40) for n in ('graph_conv', 'affine'))
    items = [(b'', b'h')] + [(k, k[::-1] * (1 + i % 3)) for i, k in enumerate(keys)]
    for bs in (64, 700, tc._BLOCK_SIZE):                   # many blocks / a few / one
        fn = str(tmp_path / ('t%d.index' % bs))
        tc.write_table(fn, items, block_size=bs)
        assert tc.read_table(fn) == items
    # sharing prefixes keeps the single-block table well below the raw key bytes
    assert os.path.getsize(fn) < sum(len(v) for _, v in items) + 0.5 * sum(len(k) for k, _ in items)
    with pytest.raises(tc.CheckpointError):
        tc.write_table(str(tmp_path / 'bad.index'), [(b'b', b''), (b'a', b'')])


def test_entry_proto_bytes():
    # BundleEntryProto{dtype: DT_FLOAT, shape{dim{size:6} dim{size:64}}, offset: 300, size: 1536, crc32c: 0x01020304}
    want = bytes.fromhex('0801' '1208' '12020806' '12020840' '20ac02' '28800c' '3504030201'.replace(' ', ''))
    got = tc.encode_entry(1, (6, 64), 0, 300, 1536, 0x01020304)
    assert got == want
    e = tc.decode_entry(want)
    assert (e['dtype'], e['shape'], e['shard_id'], e['offset'], e['size'], e['crc32c']) == (1, (6, 64), 0, 300, 1536, 0x01020304)
    # scalar: empty shape message; header: num_shards 1, little endian, version{producer 1}
    assert tc.decode_entry(tc.encode_entry(3, (), 0, 0, 4, 5))['shape'] == ()
    assert tc.encode_header(1) == bytes.fromhex('0801' '1a020801')
    assert tc.decode_header(bytes.fromhex('0802' '1a020801'))['num_shards'] == 2


def test_bundle_round_trip_and_state_file(tmp_path):
    rng = np.random.default_rng(3)
    arrays = {'generator/encoder/encoder_conv1/weights': rng.standard_normal((6, 64)).astype(np.float32),
              'generator/encoder/encoder_conv1/bias': np.full((1, 1, 64), 0.1, np.float32),
              'generator/encoder/fc_mean/dense/kernel': rng.standard_normal((5516, 64)).astype(np.float32),
              'generator/encoder/encoder_conv1/weights/Momentum': np.zeros((6, 64), np.float32),
              'training/global_step': np.asarray(1234, dtype=np.int32),
              'empty': np.zeros((0, 3), np.float32), 'f64': rng.standard_normal(7), 'flags': np.array([True, False]),
              'i64': np.arange(5, dtype=np.int64)}
    d = str(tmp_path / 'checkpoints' / 'exp')
    prefix = tc.write_bundle(os.path.join(d, 'model-1234'), arrays)
    assert sorted(os.listdir(d)) == ['model-1234.data-00000-of-00001', 'model-1234.index']
    r = tc.BundleReader(prefix)
    assert r.keys() == sorted(arrays) and r.header['num_shards'] == 1
    assert r.variable_to_shape_map()['generator/encoder/encoder_conv1/bias'] == [1, 1, 64]
    for k, a in arrays.items():
        b = r.get_tensor(k)
        assert b.dtype == a.dtype and b.shape == a.shape and np.array_equal(a, b), k
    # tensors are packed back to back in name order
    offs = [r.entries[k]['offset'] for k in r.keys()]
    sizes = [r.entries[k]['size'] for k in r.keys()]
    assert offs == list(np.cumsum([0] + sizes[:-1]))
    with pytest.raises(KeyError):
        r.get_tensor('nope')

    # state file + max_to_keep
    assert tc.latest_checkpoint(d) is None
    for step in (1234, 1300, 1400):
        p2 = tc.write_bundle(os.path.join(d, 'model-%d' % step), {'x': np.float32(step)})
        tc.update_checkpoint_state(d, p2, keep=2)
    assert tc.latest_checkpoint(d) == os.path.join(d, 'model-1400')
    assert not os.path.exists(os.path.join(d, 'model-1234.index')) and os.path.exists(os.path.join(d, 'model-1300.index'))
    assert open(os.path.join(d, 'checkpoint')).read().splitlines() == [
        'model_checkpoint_path: "model-1400"', 'all_model_checkpoint_paths: "model-1300"',
        'all_model_checkpoint_paths: "model-1400"']

    # corrupted tensor bytes are detected by the per-tensor checksum
    fn = os.path.join(d, 'model-1400.data-00000-of-00001')
    open(fn, 'wb').write(b'\x00\x00\x00\x01')
    with pytest.raises(tc.CheckpointError, match="checksum"):
        tc.BundleReader(os.path.join(d, 'model-1400')).get_tensor('x')
    assert tc.BundleReader(os.path.join(d, 'model-1400'), verify=False).get_tensor('x') != np.float32(1400)


def test_read_all_skips_string_and_sliced_entries(tmp_path):
    """A checkpoint that carries a non-numeric entry (newer TF 1.x savers add the DT_STRING tensor
    ``_CHECKPOINTABLE_OBJECT_GRAPH``) must still yield its float variables: read_all() skips what get_tensor()
    cannot decode, get_tensor() on such an entry still raises, and a names filter restricts the result."""
    w = np.arange(12, dtype=np.float32).reshape(3, 4)
    step = np.asarray(7, dtype=np.int64)
    blob = b'\x0a\x04root'                                  # some serialized proto stored as ONE string element
    str_payload = bytes([len(blob)]) + blob                   # (length varints, then the bytes; never decoded here)
    prefix = str(tmp_path / 'model-7')
    data = w.tobytes() + step.tobytes() + str_payload
    open(prefix + '.data-00000-of-00001', 'wb').write(data)
    items = [(b'', tc.encode_header(1)),
             (b'_CHECKPOINTABLE_OBJECT_GRAPH', tc.encode_entry(tc.DT_STRING, (), 0, len(w.tobytes()) + 8, len(str_payload),
                                                               tc.crc_mask(tc.crc32c(str_payload)))),
             (b'generator/decoder/outputs/weights', tc.encode_entry(1, w.shape, 0, 0, w.nbytes, tc.crc_mask(tc.crc32c(w.tobytes())))),
             (b'training/global_step', tc.encode_entry(9, (), 0, w.nbytes, 8, tc.crc_mask(tc.crc32c(step.tobytes()))))]
    tc.write_table(prefix + '.index', items)
    r = tc.BundleReader(prefix)
    assert r.keys() == ['_CHECKPOINTABLE_OBJECT_GRAPH', 'generator/decoder/outputs/weights', 'training/global_step']
    assert not r.readable('_CHECKPOINTABLE_OBJECT_GRAPH') and r.readable('training/global_step')
    got = r.read_all()
    assert sorted(got) == ['generator/decoder/outputs/weights', 'training/global_step']
    assert np.array_equal(got['generator/decoder/outputs/weights'], w) and int(got['training/global_step']) == 7
    assert sorted(r.read_all(names=lambda k: k.startswith('generator/'))) == ['generator/decoder/outputs/weights']
    assert sorted(r.read_all(names=['training/global_step'])) == ['training/global_step']
    with pytest.raises(tc.CheckpointError, match="string"):
        r.get_tensor('_CHECKPOINTABLE_OBJECT_GRAPH')
    with pytest.raises(tc.CheckpointError, match="string"):
        r.read_all(skip_unreadable=False)


def test_snappy_block_decoder():
    # literal "abcd", then a 1-byte-offset copy (len 8, offset 4), then a 2-byte-offset copy (len 4, offset 12)
    src = bytes([16]) + bytes([3 << 2]) + b'abcd' + bytes([((8 - 4) << 2) | 1, 4]) + bytes([((4 - 1) << 2) | 2, 12, 0])
    assert tc.snappy_uncompress(src) == b'abcd' * 4
    with pytest.raises(tc.CheckpointError):
        tc.snappy_uncompress(bytes([5]) + bytes([3 << 2]) + b'abcd')


def test_cli_lists_tensors(tmp_path, capsys):
    prefix = tc.write_bundle(str(tmp_path / 'model-1'), {'a/w': np.zeros((2, 3), np.float32)})
    tc.update_checkpoint_state(str(tmp_path), prefix)
    tc.main([str(tmp_path)])
    out = capsys.readouterr().out
    assert 'a/w' in out and '[2, 3]' in out and '1 tensors' in out


def test_property_round_trips(tmp_path):
    """Random tables (keys with long shared prefixes, empty values, every block size) and random bundles (names, shapes,
    dtypes) survive a write/read cycle; hypothesis drives the generators."""
    from hypothesis import HealthCheck, given, settings
    from hypothesis import strategies as st

    key = st.binary(min_size=1, max_size=40).map(lambda b: b"generator/decoder/" + b)
    table = st.dictionaries(key, st.binary(max_size=300), min_size=1, max_size=60)

    @settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
    @given(table, st.sampled_from([16, 100, 1000, tc._BLOCK_SIZE]))
    def tables(entries, block_size):
        items = sorted(entries.items())
        fn = str(tmp_path / "p.index")
        tc.write_table(fn, items, block_size=block_size)
        assert tc.read_table(fn) == items

    names = st.text(alphabet="abcdefghijklmnopqrstuvwxyz_/0123456789", min_size=1, max_size=30)
    dtypes = st.sampled_from([np.float32, np.float64, np.int32, np.int64, np.uint8, np.bool_, np.float16])
    shapes = st.lists(st.integers(0, 5), max_size=4).map(tuple)

    @settings(max_examples=25, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
    @given(st.dictionaries(names, st.tuples(dtypes, shapes, st.integers(0, 2 ** 31 - 1)), min_size=1, max_size=12))
    def bundles(spec):
        arrays = {}
        for name, (dt, shape, seed) in spec.items():
            a = np.random.default_rng(seed).integers(0, 200, size=shape)
            arrays[name] = a.astype(dt)
        prefix = tc.write_bundle(str(tmp_path / "b" / "model-1"), arrays)
        got = tc.BundleReader(prefix).read_all()
        assert set(got) == set(arrays)
        for k, a in arrays.items():
            assert got[k].dtype == a.dtype and got[k].shape == a.shape and np.array_equal(got[k], a)

    tables()
    bundles()
