"""weights from a Slideflow / Keras model directory without TensorFlow.

PARITY UNPINNED for the producer of these files (no TensorFlow here, no checkpoint in the reference): the
reader is pinned against the published constants of the format, hand-assembled blocks and this build's own
writer."""
import json
import os
import struct

import numpy as np
import pytest

from biscuit_amd import keras_import as K, tf_bundle as B, weights as W
from biscuit_amd.tfrecord import crc32c


def test_table_constants_and_footer(tmp_path):
    p = str(tmp_path / 't.index')
    B.write_table(p, {b'a': b'1', b'b': b'2'})
    raw = open(p, 'rb').read()
    # LevelDB / TensorFlow table magic, little endian, closes the 48-byte footer
    assert raw[-8:] == bytes.fromhex('57fb808b247547db')
    assert B.read_table(p) == {b'a': b'1', b'b': b'2'}
    # CRC-32C known answer (RFC 3720 B.4) and TensorFlow's mask
    assert crc32c(b'123456789') == 0xE3069283
    assert B._mask(0xE3069283) == ((0xE3069283 >> 15 | 0xE3069283 << 17) + 0xA282EAD8) & 0xFFFFFFFF


def test_hand_assembled_block_prefix_compression(tmp_path):
    # entries "apple"->"1", "apply"->"22" (shares "appl"), "b"->"" with restart points at 0 and at "b"
    e = bytes([0, 5, 1]) + b'apple' + b'1' + bytes([4, 1, 2]) + b'y' + b'22'
    r1 = len(e)
    e += bytes([0, 1, 0]) + b'b'
    blk = e + struct.pack('<III', 0, r1, 2)
    data = blk + b'\x00' + struct.pack('<I', B._mask(crc32c(blk + b'\x00')))
    meta = struct.pack('<I', 0) + struct.pack('<I', 1)
    moff = len(data)
    data += meta + b'\x00' + struct.pack('<I', B._mask(crc32c(meta + b'\x00')))
    handle = B._put_varint(0) + B._put_varint(len(blk))
    ient = bytes([0, 1, len(handle)]) + b'c' + handle
    iblk = ient + struct.pack('<II', 0, 1)
    ioff = len(data)
    data += iblk + b'\x00' + struct.pack('<I', B._mask(crc32c(iblk + b'\x00')))
    foot = B._put_varint(moff) + B._put_varint(len(meta)) + B._put_varint(ioff) + B._put_varint(len(iblk))
    data += foot + b'\x00' * (40 - len(foot)) + struct.pack('<Q', B.TABLE_MAGIC)
    p = tmp_path / 'h.index'
    p.write_bytes(data)
    assert B.read_table(str(p)) == {b'apple': b'1', b'apply': b'22', b'b': b''}


def test_table_many_blocks_and_corruption(tmp_path):
    items = {f'layer_with_weights-{i}/kernel/.ATTRIBUTES/VARIABLE_VALUE'.encode(): os.urandom(1 + i % 40) for i in range(500)}
    p = str(tmp_path / 'm.index')
    B.write_table(p, items, block_size=256, restart_interval=4)
    assert B.read_table(p) == items
    raw = bytearray(open(p, 'rb').read())
    raw[10] ^= 0x40
    open(p, 'wb').write(raw)
    with pytest.raises(B.BundleError, match='CRC'):
        B.read_table(p)
    assert len(B.read_table(p, verify=False)) == 500 or True     # unverified read may or may not parse; must not hang
    raw[-1] ^= 1
    open(p, 'wb').write(raw)
    with pytest.raises(B.BundleError, match='magic'):
        B.read_table(p)
    open(p, 'wb').write(b'short')
    with pytest.raises(B.BundleError, match='too short'):
        B.read_table(p)


def test_snappy_block_is_reported(tmp_path):
    p = str(tmp_path / 's.index')
    B.write_table(p, {b'k': b'v'})
    raw = bytearray(open(p, 'rb').read())
    blk_len = raw.index(b'\x00' * 1, 0)            # not needed: patch the first block's type byte instead
    tab = B.read_table(p)
    assert tab == {b'k': b'v'}
    # first block: entries + 8 bytes of restart array; its type byte follows
    first = B._build_block([(b'k', b'v')])
    raw[len(first)] = 1
    raw[len(first) + 1:len(first) + 5] = struct.pack('<I', B._mask(crc32c(first + b'\x01')))
    open(p, 'wb').write(raw)
    with pytest.raises(B.BundleError, match='snappy'):
        B.read_table(p)


def test_bundle_dtypes_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    t = {'f32': rng.normal(size=(3, 4, 5)).astype(np.float32), 'f16': rng.normal(size=7).astype(np.float16),
         'i64': np.asarray(123456789012, np.int64), 'i32': np.arange(6, dtype=np.int32).reshape(2, 3),
         'flag': np.asarray([True, False]), 'empty': np.zeros((0, 4), np.float32), 'graph': b'\x0a\x00hello'}
    prefix = str(tmp_path / 'variables' / 'variables')
    B.write_bundle(prefix, t, block_size=64)
    for where in (prefix, str(tmp_path / 'variables'), str(tmp_path)):
        r = B.BundleReader(where)
        assert sorted(r.keys()) == sorted(t)
        for k, v in t.items():
            got = r.tensor(k)
            if isinstance(v, bytes):
                assert got.shape == () and got.reshape(-1)[0] == v
            else:
                assert got.dtype == v.dtype and got.shape == v.shape and np.array_equal(got, v)
    # a flipped data byte is caught by the per-tensor checksum
    d = prefix + '.data-00000-of-00001'
    raw = bytearray(open(d, 'rb').read())
    off = B.BundleReader(prefix).entries['f32']['offset']
    raw[off + 5] ^= 0x10
    open(d, 'wb').write(raw)
    with pytest.raises(B.BundleError, match='CRC'):
        B.BundleReader(prefix).tensor('f32')
    assert B.BundleReader(prefix, verify=False).tensor('f32').shape == (3, 4, 5)


def test_bfloat16_entry_is_widened(tmp_path):
    prefix = str(tmp_path / 'v')
    vals = np.asarray([1.0, -2.5, 3.140625], np.float32)
    bits = (vals.view(np.uint32) >> 16).astype('<u2')
    B.write_bundle(prefix, {'x': bits})
    # patch the entry's dtype enum from DT_UINT16 (17) to DT_BFLOAT16 (14)
    tab = B.read_table(prefix + '.index')
    ent = bytearray(tab[b'x'])
    assert ent[0] == 0x08 and ent[1] == 17
    ent[1] = B.DT_BFLOAT16
    tab[b'x'] = bytes(ent)
    B.write_table(prefix + '.index', tab)
    assert np.array_equal(B.BundleReader(prefix).tensor('x'), vals)


def test_object_graph_parse():
    # root{children: layer_with_weights-0 -> 1}; node1{attribute VARIABLE_VALUE -> key}
    ref = b'\x08\x01\x12\x14layer_with_weights-0'
    root = b'\x0a' + bytes([len(ref)]) + ref
    key = b'layer_with_weights-0/kernel/.ATTRIBUTES/VARIABLE_VALUE'
    attr = b'\x0a\x0eVARIABLE_VALUE\x1a' + bytes([len(key)]) + key
    node1 = b'\x12' + bytes([len(attr)]) + attr
    g = b'\x0a' + bytes([len(root)]) + root + b'\x0a' + bytes([len(node1)]) + node1
    nodes = B.parse_object_graph(g)
    assert nodes[0]['children'] == {'layer_with_weights-0': 1}
    assert nodes[1]['attributes'] == {'VARIABLE_VALUE': key.decode()}


@pytest.fixture(scope='module')
def synth():
    return W.synthetic_weights(3)


@pytest.mark.parametrize('nested', [True, False])
def test_keras_checkpoint_round_trip(tmp_path, synth, nested):
    K.export_bundle(str(tmp_path / 'variables' / 'variables'), synth, nested=nested, optimizer_slots=True)
    w = K.from_bundle(str(tmp_path))
    assert set(w) == set(synth)
    assert all(np.array_equal(w[k], synth[k]) for k in synth)
    # and therefore the same device blob
    assert W.pack_blob(w, 'bf16') == W.pack_blob(synth, 'bf16')
    assert K.load_weights(str(tmp_path)).keys() == synth.keys()


_TENSORS = {}


def _rewrite(tmp_path, synth, edit):
    """A model directory whose checkpoint is Keras-ordered `synth` after `edit(tensors)`."""
    prefix = str(tmp_path / 'variables' / 'variables')
    if not _TENSORS:
        K.export_bundle(prefix, synth)
        r = B.BundleReader(prefix)
        _TENSORS.update({k: (r.tensor(k) if k != '_CHECKPOINTABLE_OBJECT_GRAPH' else b'') for k in r.keys()})
    t = dict(_TENSORS)
    edit(t)
    B.write_bundle(prefix, t)
    return str(tmp_path)


def test_keras_import_rejects_what_it_cannot_place(tmp_path, synth):
    sfx = '/.ATTRIBUTES/VARIABLE_VALUE'

    def drop_layer(t):
        for k in [k for k in t if k.startswith('layer_with_weights-0/layer_with_weights-4/')]:
            del t[k]
    with pytest.raises(K.ImportError_, match='separable'):
        K.from_bundle(_rewrite(tmp_path / 'a', synth, drop_layer))

    def wrong_shape(t):
        k = 'layer_with_weights-0/layer_with_weights-0/kernel' + sfx
        t[k] = np.zeros((3, 3, 3, 16), np.float32)
    with pytest.raises(K.ImportError_, match='wrong shape'):
        K.from_bundle(_rewrite(tmp_path / 'b', synth, wrong_shape))

    def third_hidden(t):
        t['layer_with_weights-9/kernel' + sfx] = np.zeros((1024, 1024), np.float32)
        t['layer_with_weights-9/bias' + sfx] = np.zeros(1024, np.float32)
    with pytest.raises(K.ImportError_, match='dense'):
        K.from_bundle(_rewrite(tmp_path / 'c', synth, third_hidden))

    def two_bn_in_a_row(t):
        # swap block2_sepconv2_bn (index 7 in Keras order) with the residual conv (index 8): [sep, conv, bn, bn]
        a, b = 'layer_with_weights-0/layer_with_weights-8/', 'layer_with_weights-0/layer_with_weights-7/'
        conv = {k: t.pop(k) for k in list(t) if k.startswith(a)}
        bn = {k: t.pop(k) for k in list(t) if k.startswith(b)}
        for k, v in conv.items():
            t[b + k[len(a):]] = v
        for k, v in bn.items():
            t[a + k[len(b):]] = v
    with pytest.raises(K.ImportError_, match='ambiguous'):
        K.from_bundle(_rewrite(tmp_path / 'd', synth, two_bn_in_a_row))

    with pytest.raises(B.BundleError, match='no checkpoint index'):
        K.from_bundle(str(tmp_path / 'nothing'))


def test_conv_bias_is_folded_into_the_moving_mean(tmp_path, synth):
    sfx = '/.ATTRIBUTES/VARIABLE_VALUE'
    bias = np.linspace(-1, 1, 32).astype(np.float32)

    def add_bias(t):
        t['layer_with_weights-0/layer_with_weights-0/bias' + sfx] = bias
    w = K.from_bundle(_rewrite(tmp_path, synth, add_bias))
    assert np.array_equal(w['block1_conv1_bn/moving_mean'], synth['block1_conv1_bn/moving_mean'] - bias)
    s0, b0 = W.fold_bn(synth, 'block1_conv1_bn')
    s1, b1 = W.fold_bn(w, 'block1_conv1_bn')
    assert np.allclose(s0, s1) and np.allclose(b1, b0 + bias * s0, atol=1e-6)


def test_named_arrays_and_safetensors(tmp_path, synth):
    # Keras' automatic names for the residual branches, ':0' suffixes, an arbitrary output-layer name
    ren = {'block2_res_conv': 'conv2d', 'block3_res_conv': 'conv2d_1', 'block4_res_conv': 'conv2d_2',
           'block13_res_conv': 'conv2d_3', 'block2_res_bn': 'batch_normalization', 'block3_res_bn': 'batch_normalization_1',
           'block4_res_bn': 'batch_normalization_2', 'block13_res_bn': 'batch_normalization_3', 'logits': 'out-cohort'}
    named = {}
    for k, v in synth.items():
        layer, _, var = k.partition('/')
        named[f'{ren.get(layer, layer)}/{var}:0'] = v
    w = K.from_named(named)
    assert all(np.array_equal(w[k], synth[k]) for k in synth)
    p = str(tmp_path / 'w.safetensors')
    K.save_safetensors(p, synth)
    w2 = K.load_weights(p)
    assert all(np.array_equal(w2[k], synth[k]) for k in synth)
    npz = str(tmp_path / 'w.npz')
    W.save_npz(npz, synth)
    w3 = K.load_weights(npz)
    assert all(np.array_equal(w3[k], synth[k]) for k in synth)
    del named['conv2d_3/kernel:0']
    with pytest.raises(K.ImportError_):
        K.from_named(named)


def test_read_params(tmp_path):
    d = tmp_path / '00001-cohort-HP0' / 'cohort-HP0_epoch1'
    d.mkdir(parents=True)
    fit = {'target_means': [65.2, 28.6, -14.8], 'target_stds': [15.8, 9.3, 6.1]}
    params = {'norm_fit': fit, 'outcome_labels': {'0': 'adenocarcinoma', '1': 'squamous'}, 'outcomes': ['cohort'],
              'tile_px': 299, 'hp': {'model': 'xception', 'tile_px': 299, 'hidden_layers': 2, 'hidden_layer_width': 1024,
                                     'dropout': 0.1, 'normalizer': 'reinhard_fast'}}
    (d.parent / 'params.json').write_text(json.dumps(params))
    got = K.read_params(str(d))                      # found in the parent, where Slideflow also keeps a copy
    assert got['norm_fit'] == fit and got['normalizer'] == 'reinhard_fast' and got['dropout'] == 0.1
    assert got['outcome_labels']['1'] == 'squamous'
    params['hp']['hidden_layers'] = 1
    (d / 'params.json').write_text(json.dumps(params))
    with pytest.raises(K.ImportError_, match='hidden_layers'):
        K.read_params(str(d))
    assert K.read_params(str(tmp_path)) is None


def test_keras_layer_order_matches_the_architecture():
    order = K.keras_layer_order()
    names = [n for n, _ in order]
    want = set(k.split('/')[0] for k in W.expected_shapes())
    assert set(names) == want and len(names) == len(want)
    i = names.index('block2_sepconv2_bn')
    assert names[i + 1:i + 3] == ['block2_res_conv', 'block2_res_bn']       # Keras: ..., conv2d, batch_normalization
    assert names[-3:] == ['hidden_0', 'hidden_1', 'logits']


def _pb(field, wt, payload):
    """One protobuf field, assembled by hand (wire types 0 varint, 2 length-delimited, 5 fixed32)."""
    tag = B._put_varint((field << 3) | wt)
    if wt == 0:
        return tag + B._put_varint(payload)
    if wt == 2:
        return tag + B._put_varint(len(payload)) + payload
    return tag + payload


def _entry_proto(dtype, shape, shard, offset, raw, sliced=False):
    """BundleEntryProto from tensorflow/core/protobuf/tensor_bundle.proto: dtype=1, shape=2, shard_id=3, offset=4,
    size=5, crc32c=6 (fixed32, masked), slices=7 -- written here field by field, not through write_bundle."""
    dims = b''.join(_pb(2, 2, _pb(1, 0, d)) for d in shape)
    e = _pb(1, 0, dtype) + _pb(2, 2, dims) + _pb(3, 0, shard) + _pb(4, 0, offset) + _pb(5, 0, len(raw)) + \
        _pb(6, 5, struct.pack('<I', B._mask(crc32c(raw))))
    if sliced:
        e += _pb(7, 2, b'\x0a\x00')
    return e


def test_hand_assembled_multi_shard_bundle(tmp_path):
    """A checkpoint as tf.train.Saver writes it from several devices: `.data-00000-of-00002`, `.data-00001-of-00002`,
    BundleHeaderProto{num_shards=2}, entries that point into either shard at non-zero offsets.  Every byte of the
    index entries is assembled from the .proto field numbers, nothing goes through this module's writer."""
    prefix = str(tmp_path / 'ckpt')
    a = np.arange(12, dtype='<f4').reshape(3, 4)
    b = (np.arange(5, dtype='<f4') - 2).astype('<f4')
    c = np.asarray([7, -9], '<i8')
    shard0 = b'\xee' * 24 + a.tobytes()                          # tensor at offset 24 of shard 0
    shard1 = b.tobytes() + b'\x00' * 3 + c.tobytes()             # two tensors in shard 1, the second unaligned
    open(prefix + '.data-00000-of-00002', 'wb').write(shard0)
    open(prefix + '.data-00001-of-00002', 'wb').write(shard1)
    header = _pb(1, 0, 2) + _pb(2, 0, 0) + _pb(3, 2, _pb(1, 0, 1))   # num_shards=2, LITTLE endian, version{producer=1}
    table = {B.HEADER_KEY: header,
             b'layer/a': _entry_proto(1, (3, 4), 0, 24, a.tobytes()),            # DT_FLOAT = 1
             b'layer/b': _entry_proto(1, (5,), 1, 0, b.tobytes()),
             b'layer/c': _entry_proto(9, (2,), 1, len(b.tobytes()) + 3, c.tobytes())}   # DT_INT64 = 9
    B.write_table(prefix + '.index', table, block_size=48)       # several blocks: shared-prefix keys across restarts
    r = B.BundleReader(prefix)
    assert r.header['num_shards'] == 2 and sorted(r.keys()) == ['layer/a', 'layer/b', 'layer/c']
    assert np.array_equal(r.tensor('layer/a'), a) and np.array_equal(r.tensor('layer/b'), b)
    assert np.array_equal(r.tensor('layer/c'), c) and r.tensor('layer/c').dtype == np.int64
    # a missing shard is an error that names the file, never an empty tensor
    os.remove(prefix + '.data-00001-of-00002')
    with pytest.raises(B.BundleError, match='data-00001-of-00002'):
        B.BundleReader(prefix).tensor('layer/b')
    assert np.array_equal(B.BundleReader(prefix).tensor('layer/a'), a)   # the other shard still reads
    # a truncated shard
    open(prefix + '.data-00001-of-00002', 'wb').write(shard1[:10])
    with pytest.raises(B.BundleError, match='truncated'):
        B.BundleReader(prefix).tensor('layer/c')


def test_big_endian_and_sliced_entries_are_refused(tmp_path):
    prefix = str(tmp_path / 'be')
    x = np.arange(4, dtype='>f4')
    open(prefix + '.data-00000-of-00001', 'wb').write(x.tobytes())
    # BundleHeaderProto.endianness = BIG (1): tensors would be byte-swapped; refuse instead of mis-reading
    B.write_table(prefix + '.index', {B.HEADER_KEY: _pb(1, 0, 1) + _pb(2, 0, 1), b'x': _entry_proto(1, (4,), 0, 0, x.tobytes())})
    with pytest.raises(B.BundleError, match='big-endian'):
        B.BundleReader(prefix)
    # a partitioned variable (BundleEntryProto.slices present) has no bytes of its own under that key
    B.write_table(prefix + '.index', {B.HEADER_KEY: _pb(1, 0, 1), b'x': _entry_proto(1, (4,), 0, 0, x.tobytes(), sliced=True)})
    with pytest.raises(B.BundleError, match='sliced'):
        B.BundleReader(prefix).tensor('x')
    # varints longer than one byte in the entry (offset 300 = 0xAC 0x02, little-endian base-128 groups)
    assert B._put_varint(300) == b'\xac\x02'
    big = b'\x00' * 300 + x.astype('<f4').tobytes()
    open(prefix + '.data-00000-of-00001', 'wb').write(big)
    B.write_table(prefix + '.index', {B.HEADER_KEY: _pb(1, 0, 1), b'x': _entry_proto(1, (4,), 0, 300, x.astype('<f4').tobytes())})
    assert np.array_equal(B.BundleReader(prefix).tensor('x'), np.arange(4, dtype=np.float32))


def test_layers_bound_by_name_when_the_savedmodel_names_them(tmp_path, synth):
    """keras_metadata.pb + the checkpoint's object graph name every layer: the import must not depend on how Keras
    numbered ``layer_with_weights-N``.  Here each shortcut convolution is numbered BEFORE the block's last
    BatchNormalization (same shape as the shortcut's own): position-based matching would swap the two."""
    order = K.keras_layer_order()
    names = [n for n, _ in order]
    for block in (2, 3, 4, 13):
        i, j = names.index(f'block{block}_sepconv2_bn'), names.index(f'block{block}_res_conv')
        assert j == i + 1
        order[i], order[j] = order[j], order[i]
        names[i], names[j] = names[j], names[i]
    d = tmp_path / 'named'
    K.export_bundle(str(d / 'variables' / 'variables'), synth, metadata=str(d / 'keras_metadata.pb'), order=order)
    w = K.from_bundle(str(d))
    assert set(w) == set(synth) and all(np.array_equal(w[k], synth[k]) for k in synth)
    # the same files without the metadata: the position-based path sees two BatchNormalizations in a row and refuses
    os.remove(d / 'keras_metadata.pb')
    with pytest.raises(K.ImportError_, match='ambiguous|BatchNormalization'):
        K.from_bundle(str(d))
    # Keras order + metadata: still fine, and a layer name that is not Xception's is an error, not a guess
    e = tmp_path / 'plain'
    K.export_bundle(str(e / 'variables' / 'variables'), synth, metadata=str(e / 'keras_metadata.pb'))
    assert W.pack_blob(K.from_bundle(str(e)), 'bf16') == W.pack_blob(synth, 'bf16')
    recs = B.parse_saved_metadata(open(e / 'keras_metadata.pb', 'rb').read())
    assert len(recs) == 2 + len(K.keras_layer_order()) and recs[2]['metadata']['name'] == 'block1_conv1'
    for r in recs:
        if r['metadata'].get('name') == 'block7_sepconv2_bn':
            r['metadata']['name'] = 'block7_sepconv9_bn'
    open(e / 'keras_metadata.pb', 'wb').write(B.build_saved_metadata(recs))
    with pytest.raises(K.ImportError_, match='block7_sepconv9_bn'):
        K.from_bundle(str(e))


def test_variables_outside_the_layer_tree(tmp_path, synth):
    """Something that looks like model weights outside the layer tree is an error; other trackables a real SavedModel
    carries (counters, metric state) are skipped with a warning -- unless strict=True."""
    def edit(t):
        t['some_other_object/kernel/.ATTRIBUTES/VARIABLE_VALUE'] = np.zeros((3, 3), np.float32)
    with pytest.raises(K.ImportError_, match='outside the layer_with_weights'):
        K.from_bundle(_rewrite(tmp_path, synth, edit))

    def extra(t):
        t['metrics/0/total/.ATTRIBUTES/VARIABLE_VALUE'] = np.zeros((), np.float32)
        t['layer-3/step_counter/.ATTRIBUTES/VARIABLE_VALUE'] = np.zeros((), np.int64)
    path = _rewrite(tmp_path, synth, extra)
    with pytest.warns(UserWarning, match='ignoring 2 variables'):
        w = K.from_bundle(path)
    assert W.pack_blob(w, 'f16') == W.pack_blob(synth, 'f16')
    with pytest.raises(K.ImportError_, match='outside the layer_with_weights'):
        K.from_bundle(path, strict=True)


def test_mutated_bundles_fail_with_bundle_errors_only(tmp_path):
    """Bytes flipped or cut anywhere in a checkpoint's index and data files: the reader answers with tensors or with
    ``BundleError`` (CRC mismatches, truncation, corrupt entries) -- never with another exception or a huge allocation."""
    rng = np.random.default_rng(0)
    tensors = {f'layer_with_weights-{i}/kernel/.ATTRIBUTES/VARIABLE_VALUE': rng.normal(size=(3, 3, 4, 8)).astype(np.float32) for i in range(12)}
    tensors['note'] = b'text'
    prefix = str(tmp_path / 'ck')
    B.write_bundle(prefix, tensors, block_size=256)
    idx = bytearray(open(prefix + '.index', 'rb').read())
    dat = bytearray(open(prefix + '.data-00000-of-00001', 'rb').read())
    outcomes = {'ok': 0, 'refused': 0}
    for k in range(400):
        bad_i, bad_d = bytearray(idx), bytearray(dat)
        if k % 2 == 0:
            at = int(rng.integers(0, len(bad_i)))
            bad_i[at] ^= int(rng.integers(1, 256))
            if k % 10 == 0:
                bad_i = bad_i[:at]
        else:
            at = int(rng.integers(0, len(bad_d)))
            bad_d[at] ^= int(rng.integers(1, 256))
            if k % 10 == 1:
                bad_d = bad_d[:at]
        p = str(tmp_path / 'm')
        open(p + '.index', 'wb').write(bytes(bad_i))
        open(p + '.data-00000-of-00001', 'wb').write(bytes(bad_d))
        for verify in (True, False):
            try:
                r = B.BundleReader(p, verify=verify)
                for name in r.keys():
                    r.tensor(name)
                outcomes['ok'] += 1
            except B.BundleError:
                outcomes['refused'] += 1
    assert outcomes['refused'] > 300 and outcomes['ok'] > 100, outcomes      # (verify=False lets flipped tensor bytes through: that is its meaning)
