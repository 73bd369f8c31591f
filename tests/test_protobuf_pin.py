"""The protobuf LAYER of the two file formats this build reads and writes without TensorFlow -- ``tf.train.Example`` (Slideflow's
tile TFRecords) and the tensor-bundle / object-graph / keras-metadata messages of a SavedModel
(section 8f row 2) -- pinned against an independent implementation: Google's protobuf runtime (``google.protobuf``, installed here),
with the messages declared from TensorFlow's published ``.proto`` files (field numbers and types as in
tensorflow/core/example/{example,feature}.proto, tensorflow/core/protobuf/{tensor_bundle,trackable_object_graph}.proto,
tensorflow/core/framework/{tensor_shape,versions}.proto, keras/protobuf/saved_metadata.proto).

Both directions: (i) what this build WRITES is decoded by Google's runtime, (ii) what Google's runtime ENCODES -- including forms this
build's own writer never produces: packed and unpacked repeated scalars, fields in another order, unknown fields, negative int64 --
is read by this build's parsers (the Python one and the native reader of libbiscuit_io).

What this does NOT pin: the table / block / CRC layer around the messages (LevelDB's sorted-string-table format: self-checked against
its published constants in tests/test_weight_import.py) and the choice of names TensorFlow writes -- rows (c) / (f2) of SURVEY.md
section 8 stay "unpinned" until a TensorFlow-written file exists (tools/make_tf_fixture.py)."""
import json
import struct

import numpy as np
import pytest

from biscuit_amd import tf_bundle as B, tfrecord as T

pb = pytest.importorskip('google.protobuf')
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory      # noqa: E402

F = descriptor_pb2.FieldDescriptorProto


def _file():
    f = descriptor_pb2.FileDescriptorProto(name='bq_pin.proto', package='bqpin', syntax='proto3')

    def msg(name, fields, nested=()):
        m = descriptor_pb2.DescriptorProto(name=name)
        for fname, num, ftype, label, tname, extra in fields:
            fd = m.field.add(name=fname, number=num, type=ftype, label=label)
            if tname:
                fd.type_name = tname
            for k, v in (extra or {}).items():
                if k == 'oneof':
                    fd.oneof_index = v
                elif k == 'packed':
                    fd.options.packed = v
        for n in nested:
            m.nested_type.append(n)
        return m
    OPT, REP = F.LABEL_OPTIONAL, F.LABEL_REPEATED
    # ---- example.proto / feature.proto
    f.message_type.append(msg('BytesList', [('value', 1, F.TYPE_BYTES, REP, None, None)]))
    f.message_type.append(msg('FloatList', [('value', 1, F.TYPE_FLOAT, REP, None, {'packed': True})]))
    f.message_type.append(msg('Int64List', [('value', 1, F.TYPE_INT64, REP, None, {'packed': True})]))
    f.message_type.append(msg('Int64ListUnpacked', [('value', 1, F.TYPE_INT64, REP, None, {'packed': False})]))   # the proto2-era wire form
    feat = msg('Feature', [('bytes_list', 1, F.TYPE_MESSAGE, OPT, '.bqpin.BytesList', {'oneof': 0}),
                           ('float_list', 2, F.TYPE_MESSAGE, OPT, '.bqpin.FloatList', {'oneof': 0}),
                           ('int64_list', 3, F.TYPE_MESSAGE, OPT, '.bqpin.Int64List', {'oneof': 0})])
    feat.oneof_decl.add(name='kind')
    f.message_type.append(feat)
    entry = msg('FeatureEntry', [('key', 1, F.TYPE_STRING, OPT, None, None), ('value', 2, F.TYPE_MESSAGE, OPT, '.bqpin.Feature', None)])
    entry.options.map_entry = True
    f.message_type.append(msg('Features', [('feature', 1, F.TYPE_MESSAGE, REP, '.bqpin.Features.FeatureEntry', None)], nested=[entry]))
    f.message_type.append(msg('Example', [('features', 1, F.TYPE_MESSAGE, OPT, '.bqpin.Features', None)]))
    # ---- versions.proto, tensor_shape.proto, tensor_bundle.proto
    f.message_type.append(msg('VersionDef', [('producer', 1, F.TYPE_INT32, OPT, None, None), ('min_consumer', 2, F.TYPE_INT32, OPT, None, None),
                                             ('bad_consumers', 3, F.TYPE_INT32, REP, None, None)]))
    dim = msg('Dim', [('size', 1, F.TYPE_INT64, OPT, None, None), ('name', 2, F.TYPE_STRING, OPT, None, None)])
    f.message_type.append(msg('TensorShapeProto', [('dim', 2, F.TYPE_MESSAGE, REP, '.bqpin.TensorShapeProto.Dim', None),
                                                   ('unknown_rank', 3, F.TYPE_BOOL, OPT, None, None)], nested=[dim]))
    f.message_type.append(msg('BundleHeaderProto', [('num_shards', 1, F.TYPE_INT32, OPT, None, None), ('endianness', 2, F.TYPE_INT32, OPT, None, None),
                                                    ('version', 3, F.TYPE_MESSAGE, OPT, '.bqpin.VersionDef', None)]))
    f.message_type.append(msg('BundleEntryProto', [('dtype', 1, F.TYPE_INT32, OPT, None, None),
                                                   ('shape', 2, F.TYPE_MESSAGE, OPT, '.bqpin.TensorShapeProto', None),
                                                   ('shard_id', 3, F.TYPE_INT32, OPT, None, None), ('offset', 4, F.TYPE_INT64, OPT, None, None),
                                                   ('size', 5, F.TYPE_INT64, OPT, None, None), ('crc32c', 6, F.TYPE_FIXED32, OPT, None, None)]))
    # ---- trackable_object_graph.proto
    ref = msg('ObjectReference', [('node_id', 1, F.TYPE_INT32, OPT, None, None), ('local_name', 2, F.TYPE_STRING, OPT, None, None)])
    ten = msg('SerializedTensor', [('name', 1, F.TYPE_STRING, OPT, None, None), ('full_name', 2, F.TYPE_STRING, OPT, None, None),
                                   ('checkpoint_key', 3, F.TYPE_STRING, OPT, None, None)])
    slot = msg('SlotVariableReference', [('original_variable_node_id', 1, F.TYPE_INT32, OPT, None, None), ('slot_name', 2, F.TYPE_STRING, OPT, None, None),
                                         ('slot_variable_node_id', 3, F.TYPE_INT32, OPT, None, None)])
    obj = msg('TrackableObject', [('children', 1, F.TYPE_MESSAGE, REP, '.bqpin.TrackableObjectGraph.TrackableObject.ObjectReference', None),
                                  ('attributes', 2, F.TYPE_MESSAGE, REP, '.bqpin.TrackableObjectGraph.TrackableObject.SerializedTensor', None),
                                  ('slot_variables', 3, F.TYPE_MESSAGE, REP, '.bqpin.TrackableObjectGraph.TrackableObject.SlotVariableReference', None)],
              nested=[ref, ten, slot])
    f.message_type.append(msg('TrackableObjectGraph', [('nodes', 1, F.TYPE_MESSAGE, REP, '.bqpin.TrackableObjectGraph.TrackableObject', None)],
                              nested=[obj]))
    # ---- saved_metadata.proto (keras)
    f.message_type.append(msg('SavedObject', [('node_id', 2, F.TYPE_INT32, OPT, None, None), ('node_path', 3, F.TYPE_STRING, OPT, None, None),
                                              ('identifier', 4, F.TYPE_STRING, OPT, None, None), ('metadata', 5, F.TYPE_STRING, OPT, None, None),
                                              ('version', 6, F.TYPE_MESSAGE, OPT, '.bqpin.VersionDef', None)]))
    f.message_type.append(msg('SavedMetadata', [('nodes', 1, F.TYPE_MESSAGE, REP, '.bqpin.SavedObject', None)]))
    return f


@pytest.fixture(scope='module')
def M():
    pool = descriptor_pool.DescriptorPool()
    pool.Add(_file())
    names = ['Example', 'Features', 'Feature', 'BytesList', 'Int64List', 'Int64ListUnpacked', 'FloatList', 'BundleHeaderProto',
             'BundleEntryProto', 'TensorShapeProto', 'TrackableObjectGraph', 'SavedMetadata', 'VersionDef']
    return {n: message_factory.GetMessageClass(pool.FindMessageTypeByName('bqpin.' + n)) for n in names}


# ------------------------------------------------------------------------------------------------ tf.train.Example
def test_examples_this_build_writes_are_decoded_by_googles_runtime(M):
    png = T.encode_image(np.random.default_rng(0).integers(0, 256, (299, 299, 3), dtype=np.uint8))
    for slide, x, y in (('TCGA-05-4244', 1234, 99999), ('s', 0, 0), ('ünï', 2 ** 40, 7)):
        ex = M['Example']()
        ex.ParseFromString(T.encode_example(slide, png, x, y))
        f = ex.features.feature
        assert set(f) == {'slide', 'image_raw', 'loc_x', 'loc_y'}
        assert f['slide'].bytes_list.value == [slide.encode()] and f['image_raw'].bytes_list.value == [png]
        assert list(f['loc_x'].int64_list.value) == [x] and list(f['loc_y'].int64_list.value) == [y]
        assert f['slide'].WhichOneof('kind') == 'bytes_list' and f['loc_x'].WhichOneof('kind') == 'int64_list'


def _google_example(M, slide, png, x, y, order=('image_raw', 'loc_x', 'slide', 'loc_y'), extra=False):
    ex = M['Example']()
    for k in order:                              # (map order on the wire = insertion order: another order than this build's writer)
        f = ex.features.feature[k]
        if k == 'slide':
            f.bytes_list.value.append(slide.encode())
        elif k == 'image_raw':
            f.bytes_list.value.append(png)
        else:
            f.int64_list.value.append(x if k == 'loc_x' else y)
    if extra:                                    # a feature this path does not know (Slideflow versions differ in what they store)
        ex.features.feature['tfr_index'].int64_list.value.append(17)
        ex.features.feature['mpp'].float_list.value.append(0.5)
    return ex.SerializeToString()


def test_examples_googles_runtime_encodes_are_read_here(M, tmp_path):
    from biscuit_amd import tfrecord_native as tn
    tiles = np.random.default_rng(1).integers(0, 256, (3, 299, 299, 3), dtype=np.uint8)
    pngs = [T.encode_image(t) for t in tiles]
    locs = [(5, 6), (-3, 2 ** 33), (0, 123456)]                       # (a negative int64 is ten varint bytes)
    payloads = [_google_example(M, 'sl', pngs[i], *locs[i], extra=(i == 1)) for i in range(3)]
    # ... and the unpacked wire form of a repeated int64 (what a proto2-era writer emits) for loc_x of the third record: its Feature
    # messages come from Google's runtime, the map entries around them are put together by hand
    un = M['Int64ListUnpacked'](value=[locs[2][0]]).SerializeToString()
    assert un != M['Int64List'](value=[locs[2][0]]).SerializeToString()                  # really another encoding
    feats = {'slide': M['Feature'](bytes_list=M['BytesList'](value=[b'sl'])).SerializeToString(),
             'image_raw': M['Feature'](bytes_list=M['BytesList'](value=[pngs[2]])).SerializeToString(),
             'loc_x': T._ld(3, un),
             'loc_y': M['Feature'](int64_list=M['Int64List'](value=[locs[2][1]])).SerializeToString()}
    payloads[2] = T._ld(1, b''.join(T._ld(1, T._ld(1, k.encode()) + T._ld(2, v)) for k, v in feats.items()))
    chk = M['Example']()
    chk.ParseFromString(payloads[2])                                                      # (Google's runtime accepts both forms too)
    assert list(chk.features.feature['loc_x'].int64_list.value) == [locs[2][0]]
    for p, (x, y), png in zip(payloads, locs, pngs):
        got = T.parse_example(p)
        assert got['slide'] == b'sl' and got['image_raw'] == png and (got['loc_x'], got['loc_y']) == ([x], [y])
    # the same records framed into a file: the Python reader and the native reader of libbiscuit_io
    path = str(tmp_path / 'g.tfrecords')
    with open(path, 'wb') as f:
        for p in payloads:
            hdr = struct.pack('<Q', len(p))
            f.write(hdr + struct.pack('<I', T.masked_crc(hdr)) + p + struct.pack('<I', T.masked_crc(p)))
    name, got, loc = T.read_slide(path, 299, native=False)
    assert name == 'sl' and np.array_equal(got, tiles) and loc.tolist() == [list(v) for v in locs]
    if tn.available():
        with tn.NativeReader(path, verify='full') as r:
            assert r.slide == 'sl' and len(r) == 3
            d, l2 = r.decode(0, 3, 299)
            assert np.array_equal(d, tiles) and l2.tolist() == [list(v) for v in locs]


# ------------------------------------------------------------------------------------------------ tensor bundle
def test_bundle_this_build_writes_is_decoded_by_googles_runtime(M, tmp_path):
    rng = np.random.default_rng(2)
    tensors = {'a/kernel': rng.normal(size=(3, 3, 8, 1)).astype(np.float32), 'b/bias': rng.normal(size=(5,)).astype(np.float32),
               'c/step': np.array(7, np.int64), 'note': b'hello'}
    prefix = str(tmp_path / 'ck')
    B.write_bundle(prefix, tensors)
    table = B.read_table(prefix + '.index')
    hdr = M['BundleHeaderProto']()
    hdr.ParseFromString(table[b''])
    assert hdr.num_shards == 1 and hdr.endianness == 0 and hdr.version.producer == 1
    data = open(prefix + '.data-00000-of-00001', 'rb').read()
    for name, want in tensors.items():
        e = M['BundleEntryProto']()
        e.ParseFromString(table[name.encode()])
        assert e.shard_id == 0 and e.crc32c == B._mask(T.crc32c(data[e.offset:e.offset + e.size]))
        if isinstance(want, bytes):
            assert e.dtype == B.DT_STRING and len(e.shape.dim) == 0
        else:
            assert e.dtype == B._DT_OF[want.dtype] and [d.size for d in e.shape.dim] == list(want.shape)
            assert np.array_equal(np.frombuffer(data[e.offset:e.offset + e.size], want.dtype).reshape(want.shape), want)


def test_bundle_entries_googles_runtime_encodes_are_read_here(M, tmp_path):
    rng = np.random.default_rng(3)
    tensors = {'x/kernel': rng.normal(size=(2, 3, 4)).astype(np.float32), 'y/count': np.arange(6, dtype=np.int64).reshape(2, 3),
               'z/half': rng.normal(size=(7,)).astype(np.float16)}
    data, items = bytearray(), {}
    hdr = M['BundleHeaderProto'](num_shards=1, endianness=0)
    hdr.version.producer = 1
    hdr.version.bad_consumers.extend([3, 5])                   # (a field this build's parser has to skip)
    items[b''] = hdr.SerializeToString()
    for name in sorted(tensors):
        a = tensors[name]
        raw = a.tobytes()
        e = M['BundleEntryProto'](dtype=B._DT_OF[a.dtype], shard_id=0, offset=len(data), size=len(raw), crc32c=B._mask(T.crc32c(raw)))
        for d in a.shape:
            e.shape.dim.add(size=d, name='')
        items[name.encode()] = e.SerializeToString()
        data += raw
    prefix = str(tmp_path / 'g')
    open(prefix + '.data-00000-of-00001', 'wb').write(data)
    B.write_table(prefix + '.index', items)                     # (the table layer is this build's: not what is pinned here)
    r = B.BundleReader(prefix)
    assert sorted(r.keys()) == sorted(tensors) and r.header == {'num_shards': 1, 'endianness': 0}
    for name, a in tensors.items():
        assert r.shape(name) == a.shape and np.array_equal(r.tensor(name), a)


def test_object_graph_and_keras_metadata_both_ways(M):
    nodes = [{'children': {'layer_with_weights-0': 1, 'optimizer': 2}, 'attributes': {}},
             {'children': {}, 'attributes': {'kernel': 'layer_with_weights-0/kernel/.ATTRIBUTES/VARIABLE_VALUE',
                                             'bias': 'layer_with_weights-0/bias/.ATTRIBUTES/VARIABLE_VALUE'}},
             {'children': {}, 'attributes': {}}]
    g = M['TrackableObjectGraph']()
    g.ParseFromString(B.build_object_graph(nodes))                                   # this build's bytes -> Google's runtime
    assert len(g.nodes) == 3 and {c.local_name: c.node_id for c in g.nodes[0].children} == nodes[0]['children']
    assert {a.name: a.checkpoint_key for a in g.nodes[1].attributes} == nodes[1]['attributes']
    g.nodes[1].attributes[0].full_name = 'dense/kernel'                              # fields this build's writer never emits
    g.nodes[2].slot_variables.add(original_variable_node_id=1, slot_name='m', slot_variable_node_id=2)
    assert B.parse_object_graph(g.SerializeToString()) == nodes                      # Google's bytes -> this build's parser
    recs = [{'node_id': 4, 'node_path': 'root.layer_with_weights-0', 'identifier': '_tf_keras_layer',
             'metadata': {'name': 'block1_conv1', 'class_name': 'Conv2D', 'config': {'filters': 32}}}]
    sm = M['SavedMetadata']()
    sm.ParseFromString(B.build_saved_metadata(recs))
    assert sm.nodes[0].node_id == 4 and sm.nodes[0].node_path == recs[0]['node_path'] and json.loads(sm.nodes[0].metadata) == recs[0]['metadata']
    sm.nodes[0].version.producer = 2
    assert B.parse_saved_metadata(sm.SerializeToString()) == recs
