"""Weight import from a Slideflow / Keras model directory.

The reference evaluates a trained model found by ``biscuit/utils.py:233-272`` (``find_model``: the directory
``{models_dir}/NNNNN-{outcome}-{label}-HP0/{outcome}-{label}-HP0_epoch1``) through
``Project.evaluate(model=...)`` (``biscuit/experiment.py:912-922``); Slideflow writes it as a Keras
SavedModel (``saved_model.pb`` + ``variables/``) next to a ``params.json`` that carries the stain
normaliser fit (``norm_fit``), the hyper-parameters (``hp``, cf. ``biscuit/hp.py:3-23``) and the outcome
labels.  ``load_model_dir`` turns that directory into the canonical weight dict of ``biscuit_amd.weights``
(float32 arrays under Keras variable names) without TensorFlow:

  * variables are read with ``tf_bundle.BundleReader``; optimizer slots, metrics and counters are skipped;
  * object-based checkpoints name variables by their path from the model object
    (``layer_with_weights-0/layer_with_weights-7/depthwise_kernel/.ATTRIBUTES/VARIABLE_VALUE``), not by
    layer name.  When the SavedModel carries ``keras_metadata.pb`` (TF >= 2.5) and the checkpoint its object graph,
    every layer is bound BY NAME: the metadata gives the Keras name of each object path, the object graph says which
    object a ``layer_with_weights-N`` path is (``block5_sepconv2_bn`` is that layer wherever Keras numbers it; the
    automatically named ``conv2d[_k]`` / ``batch_normalization[_k]`` of the four shortcuts are taken in the order of
    their suffixes).  Without it, layers are put in order by those indices (nested models expand in place) and matched
    to the architecture of keras.applications.Xception + Slideflow's two hidden layers by kind and shape:
    the 3x3 convolutions are ``block1_conv1/2``, the separable convolutions are ``sepconv_plan()`` in order,
    the 1x1 convolutions the four residual branches in order, the dense layers ``hidden_0``, ``hidden_1``,
    ``logits``; a BatchNormalization belongs to the closest preceding convolution that has none yet
    (Keras lists ``block2_sepconv2_bn, conv2d, block2_pool, batch_normalization``).  Every shape is checked;
    anything unexpected is an error, never a guess.

PARITY UNPINNED: no TensorFlow-written model exists in the container or the reference repository; the tests
round-trip synthetic weights through this module's own writer in the Keras order described above.
"""
import json
import os
import re

import numpy as np

from . import tf_bundle, weights as W

_SUFFIX = '/.ATTRIBUTES/VARIABLE_VALUE'
_BN_VARS = ('gamma', 'beta', 'moving_mean', 'moving_variance')


class ImportError_(ValueError):
    """The directory does not hold the hp.nature2022 classifier (or not in a layout this importer knows)."""


def keras_layer_order(n_hidden=2):
    """[(canonical layer name, kind)] of the layers with weights, in the order Keras lists them
    (``model.layers``: by depth, so a residual 1x1 conv sits between the block's last BN and its own BN)."""
    order = [('block1_conv1', 'conv'), ('block1_conv1_bn', 'bn'), ('block1_conv2', 'conv'), ('block1_conv2_bn', 'bn')]

    def sep(block, n):
        for i in range(1, n + 1):
            order.append((f'block{block}_sepconv{i}', 'sep'))
            order.append((f'block{block}_sepconv{i}_bn', 'bn'))

    def res(block):
        order.append((f'block{block}_res_conv', 'conv'))
        order.append((f'block{block}_res_bn', 'bn'))
    for block in (2, 3, 4):
        sep(block, 2)
        res(block)
    for block in range(5, 13):
        sep(block, 3)
    sep(13, 2)
    res(13)
    sep(14, 2)
    order += [(f'hidden_{i}', 'dense') for i in range(n_hidden)] + [('logits', 'dense')]
    return order


def _layer_key(path):
    """('layer_with_weights-0', 'layer_with_weights-12') -> (0, 12); None if a component is not indexed."""
    out = []
    for comp in path:
        m = re.fullmatch(r'layer_with_weights-(\d+)', comp)
        if not m:
            return None
        out.append(int(m.group(1)))
    return tuple(out)


def _collect_layers(reader):
    layers = {}
    skipped = []
    for key in reader.keys():
        if not key.endswith(_SUFFIX) or '.OPTIMIZER_SLOT' in key:
            continue
        path = key[:-len(_SUFFIX)].split('/')
        if path[0] in ('optimizer', 'keras_api', 'save_counter') or len(path) < 2:
            continue
        lk = _layer_key(path[:-1])
        if lk is None:
            skipped.append(key)
            continue
        layers.setdefault(lk, {})[path[-1]] = key
    if not layers:
        raise ImportError_('no layer_with_weights-N/... variables in the checkpoint '
                           f'(keys look like {reader.keys()[:3]}): not an object-based Keras checkpoint')
    order = sorted(layers)
    return [layers[k] for k in order], order, skipped


def _kind(vars_, reader):
    names = set(vars_)
    if names >= set(_BN_VARS) and names <= set(_BN_VARS):
        return 'bn'
    if 'depthwise_kernel' in names and 'pointwise_kernel' in names and names <= {'depthwise_kernel', 'pointwise_kernel', 'bias'}:
        return 'sep'
    if 'kernel' in names and names <= {'kernel', 'bias'}:
        nd = len(reader.shape(vars_['kernel']))
        return 'conv' if nd == 4 else 'dense' if nd == 2 else None
    return None


def _keras_names(reader, prefix):
    """{layer_with_weights index path: Keras layer name} from ``keras_metadata.pb`` + the checkpoint's object graph, or
    None when either is missing."""
    cands = [os.path.join(prefix, 'keras_metadata.pb'), os.path.join(os.path.dirname(os.path.abspath(prefix)), 'keras_metadata.pb'),
             os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(prefix))), 'keras_metadata.pb')]
    meta = next((c for c in cands if os.path.isfile(c)), None)
    graph = reader.object_graph()
    if meta is None or not graph:
        return None
    nodes = tf_bundle.parse_object_graph(graph)
    if not nodes:
        return None
    name_of_path = {r['node_path']: r['metadata'].get('name') for r in tf_bundle.parse_saved_metadata(open(meta, 'rb').read())
                    if isinstance(r['metadata'], dict) and r['metadata'].get('name')}
    # every path from the root through layer-K / layer_with_weights-N children, per node
    name_of_node, weighted = {}, {}
    stack, seen = [(0, 'root', ())], set()
    while stack:
        nid, path, wpath = stack.pop()
        if path in name_of_path:
            name_of_node.setdefault(nid, name_of_path[path])
        if wpath is not None and wpath:
            weighted[wpath] = nid
        if (nid, wpath) in seen or nid >= len(nodes):
            continue
        seen.add((nid, wpath))
        for child, cid in nodes[nid]['children'].items():
            m = re.fullmatch(r'layer_with_weights-(\d+)', child)
            if m:
                stack.append((cid, f'{path}.{child}', None if wpath is None else wpath + (int(m.group(1)),)))
            elif re.fullmatch(r'layer-\d+', child):
                stack.append((cid, f'{path}.{child}', None))
    names = {wp: name_of_node.get(nid) for wp, nid in weighted.items()}
    return names if all(names.values()) and names else None


def _bind_by_name(layers, keys, kinds, names, reader):
    """{layer index: canonical layer name} for convolutions, separable convolutions and BatchNormalizations from their
    Keras names; the dense layers keep their order."""
    def suffix(n, base):
        return int(n[len(base) + 1:] or 0) if n != base else 0
    out = {}
    auto_conv = sorted((suffix(names[k], 'conv2d'), i) for i, k in enumerate(keys) if re.fullmatch(r'conv2d(_\d+)?', names[k]))
    auto_bn = sorted((suffix(names[k], 'batch_normalization'), i) for i, k in enumerate(keys)
                     if re.fullmatch(r'batch_normalization(_\d+)?', names[k]))
    plan_r = W.residual_plan()
    if len(auto_conv) != len(plan_r) or len(auto_bn) != len(plan_r):
        raise ImportError_(f'{len(auto_conv)} conv2d / {len(auto_bn)} batch_normalization layers with automatic names; '
                           f'the four shortcuts of keras.applications.Xception have one of each')
    for n, (_, i) in enumerate(auto_conv):
        out[i] = plan_r[n][0] + '_conv'
    for n, (_, i) in enumerate(auto_bn):
        out[i] = plan_r[n][0] + '_bn'
    want = {name for name, kind in keras_layer_order() if kind != 'dense'}
    for i, k in enumerate(keys):
        if i in out or kinds[i] == 'dense':
            continue
        if names[k] not in want:
            raise ImportError_(f'layer {names[k]!r} ({kinds[i]}) is not a layer of keras.applications.Xception')
        out[i] = names[k]
    if len(set(out.values())) != len(out):
        raise ImportError_('two layers of the checkpoint carry the same Keras name')
    return out


_WEIGHT_LEAVES = ('kernel', 'depthwise_kernel', 'pointwise_kernel', 'bias') + tuple(_BN_VARS)


def from_bundle(prefix, verify=True, strict=False):
    """Canonical weight dict from a checkpoint prefix / ``variables`` directory / SavedModel directory.

    Variables outside the ``layer_with_weights-N`` tree: a real Keras / Slideflow SavedModel can carry extra trackables
    (metrics, step counters, custom attributes), so those are skipped with a warning -- unless one LOOKS like model weights
    (a floating-point ``kernel`` / ``gamma`` / ``moving_*`` ... leaf), which would mean layers this importer did not see:
    that, or any skipped variable with ``strict=True``, is an error."""
    reader = tf_bundle.BundleReader(prefix, verify=verify)
    layers, keys, skipped = _collect_layers(reader)
    if skipped:
        def weight_like(key):
            leaf = key[:-len(_SUFFIX)].split('/')[-1]
            floating = reader.entries[key]['dtype'] in (1, 2, 14, 19)     # DT_FLOAT, DT_DOUBLE, DT_BFLOAT16, DT_HALF
            return leaf in _WEIGHT_LEAVES and floating
        suspicious = [k for k in skipped if weight_like(k)]
        if suspicious or strict:
            bad = suspicious or skipped
            raise ImportError_(f'{len(bad)} variables outside the layer_with_weights-N tree (first: {bad[0]}): '
                               'not the hp.nature2022 classifier, or a layout this importer does not know')
        import warnings
        warnings.warn(f'keras_import: ignoring {len(skipped)} variables outside the layer tree (first: {skipped[0]})')
    kinds = []
    for n, lv in enumerate(layers):
        k = _kind(lv, reader)
        if k is None:
            raise ImportError_(f'layer {n} has variables {sorted(lv)}: not a layer of the Xception classifier')
        kinds.append(k)
    names = _keras_names(reader, prefix)
    if names is not None and all(k in names for k in keys):
        return _assemble_named(layers, kinds, _bind_by_name(layers, keys, kinds, names, reader), reader)
    convs3 = [i for i, (k, lv) in enumerate(zip(kinds, layers)) if k == 'conv' and reader.shape(lv['kernel'])[:2] == (3, 3)]
    convs1 = [i for i, (k, lv) in enumerate(zip(kinds, layers)) if k == 'conv' and reader.shape(lv['kernel'])[:2] == (1, 1)]
    seps = [i for i, k in enumerate(kinds) if k == 'sep']
    dense = [i for i, k in enumerate(kinds) if k == 'dense']
    plan_s, plan_r = W.sepconv_plan(), W.residual_plan()
    if len(convs3) != 2 or len(convs1) != len(plan_r) or len(seps) != len(plan_s) or len(convs3) + len(convs1) != kinds.count('conv'):
        raise ImportError_(f'found {len(convs3)} 3x3 convolutions, {len(convs1)} 1x1 convolutions, {len(seps)} separable '
                           f'convolutions; keras.applications.Xception has 2, {len(plan_r)} and {len(plan_s)}')
    if len(dense) != 3:
        raise ImportError_(f'found {len(dense)} dense layers; hp.nature2022 (biscuit/hp.py:13,21) has hidden_layers=2 + logits')
    name_of = {convs3[0]: 'block1_conv1', convs3[1]: 'block1_conv2'}
    name_of.update({i: plan_r[n][0] + '_conv' for n, i in enumerate(convs1)})
    name_of.update({i: plan_s[n][0] for n, i in enumerate(seps)})
    name_of.update({dense[0]: 'hidden_0', dense[1]: 'hidden_1', dense[2]: 'logits'})
    # BatchNormalization -> the closest preceding convolution without one; two in a row cannot be told apart
    bn_of, open_convs = {}, []
    prev_bn = False
    for i, k in enumerate(kinds):
        if k in ('conv', 'sep'):
            open_convs.append(i)
            prev_bn = False
        elif k == 'bn':
            if not open_convs:
                raise ImportError_(f'layer {i}: BatchNormalization with no convolution in front of it')
            if prev_bn and len(open_convs) > 0:
                raise ImportError_(f'layers {i - 1},{i}: two BatchNormalization layers in a row, ambiguous layer order')
            bn_of[open_convs.pop()] = i
            prev_bn = True
        else:
            prev_bn = False
    if open_convs:
        raise ImportError_(f'{name_of[open_convs[0]]}: no BatchNormalization found for it')

    def bn_name(conv):
        return conv[:-len('_conv')] + '_bn' if conv.endswith('_res_conv') else conv + '_bn'
    w = {}
    for i, name in name_of.items():
        lv = layers[i]
        for var, key in lv.items():
            if var != 'bias' or kinds[i] == 'dense':
                w[f'{name}/{var}'] = reader.tensor(key).astype(np.float32)
        if kinds[i] in ('conv', 'sep'):
            bl = layers[bn_of[i]]
            for var in _BN_VARS:
                w[f'{bn_name(name)}/{var}'] = reader.tensor(bl[var]).astype(np.float32)
            if 'bias' in lv:        # y = BN(conv + b): fold the bias into the moving mean
                w[f'{bn_name(name)}/moving_mean'] = w[f'{bn_name(name)}/moving_mean'] - reader.tensor(lv['bias']).astype(np.float32)
    n_classes = int(w['logits/kernel'].shape[-1]) if w['logits/kernel'].ndim == 2 else -1
    try:
        return W.validate(w, n_classes)
    except ValueError as e:
        raise ImportError_(str(e)) from None


def _assemble_named(layers, kinds, name_of, reader):
    """Canonical dict from {layer index: canonical name} (name-bound checkpoints)."""
    dense = [i for i, k in enumerate(kinds) if k == 'dense']
    if len(dense) != 3:
        raise ImportError_(f'found {len(dense)} dense layers; hp.nature2022 (biscuit/hp.py:13,21) has hidden_layers=2 + logits')
    name_of = dict(name_of)
    name_of.update({dense[0]: 'hidden_0', dense[1]: 'hidden_1', dense[2]: 'logits'})
    w, conv_bias = {}, {}
    for i, name in name_of.items():
        for var, key in layers[i].items():
            if var == 'bias' and kinds[i] in ('conv', 'sep'):
                conv_bias[name] = reader.tensor(key).astype(np.float32)
            else:
                w[f'{name}/{var}'] = reader.tensor(key).astype(np.float32)
    for name, b in conv_bias.items():                 # y = BN(conv + b): fold the bias into the moving mean
        bn = name[:-len('_conv')] + '_bn' if name.endswith('_res_conv') else name + '_bn'
        w[f'{bn}/moving_mean'] = w[f'{bn}/moving_mean'] - b
    n_classes = int(w['logits/kernel'].shape[-1]) if w.get('logits/kernel', np.zeros(0)).ndim == 2 else -1
    try:
        return W.validate(w, n_classes)
    except ValueError as e:
        raise ImportError_(str(e)) from None


def from_named(arrays):
    """Canonical weight dict from arrays under Keras names (``model.get_layer(n).get_weights()`` exported to
    npz / safetensors): residual layers may carry Keras' automatic names (``conv2d``, ``conv2d_1`` ...,
    ``batch_normalization`` ...), variable names may end in ``:0``, the output layer may have any name."""
    def base(k):
        k = k.replace('__', '/')
        return k[:-2] if k.endswith(':0') else k
    arrays = {base(k): np.asarray(v, np.float32) for k, v in arrays.items()}

    def family(prefix):
        names = sorted({k.split('/')[0] for k in arrays if re.fullmatch(prefix + r'(_\d+)?', k.split('/')[0])},
                       key=lambda n: int(n[len(prefix) + 1:]) if len(n) > len(prefix) else 0)
        return names
    ren = {}
    plan_r = W.residual_plan()
    convs, bns = family('conv2d'), family('batch_normalization')
    if convs or bns:
        if len(convs) != len(plan_r) or len(bns) != len(plan_r):
            raise ImportError_(f'{len(convs)} conv2d* / {len(bns)} batch_normalization* layers; expected {len(plan_r)} each')
        for (name, _, _), c, b in zip(plan_r, convs, bns):
            ren[c], ren[b] = name + '_conv', name + '_bn'
    known = set(k.split('/')[0] for k in W.expected_shapes())
    others = sorted({k.split('/')[0] for k in arrays} - known - set(ren))
    if 'logits' not in {k.split('/')[0] for k in arrays} and len(others) == 1:
        ren[others[0]] = 'logits'
    w = {}
    for k, v in arrays.items():
        layer, _, var = k.partition('/')
        w[f'{ren.get(layer, layer)}/{var}'] = v
    n_classes = int(w['logits/kernel'].shape[-1]) if 'logits/kernel' in w else 2
    try:
        return W.validate(w, n_classes)
    except ValueError as e:
        raise ImportError_(str(e)) from None


def save_safetensors(path, w):
    from safetensors.numpy import save_file
    save_file({k: np.ascontiguousarray(v, np.float32) for k, v in W.validate(w, _ncls(w)).items()}, path)


def load_safetensors(path):
    from safetensors.numpy import load_file
    return from_named(load_file(path))


def _ncls(w):
    return int(np.shape(w['logits/kernel'])[-1])


def read_params(model_dir):
    """``params.json`` of a Slideflow model directory (or of its parent): the fields this path uses.
    Returns {'norm_fit', 'normalizer', 'outcome_labels', 'outcomes', 'hp', 'tile_px'} with None for what
    is absent, and raises if the hyper-parameters are not the hp.nature2022 classifier (biscuit/hp.py:3-23)."""
    for d in (model_dir, os.path.dirname(os.path.abspath(model_dir))):
        p = os.path.join(d, 'params.json')
        if os.path.isfile(p):
            with open(p) as f:
                raw = json.load(f)
            break
    else:
        return None
    hp = raw.get('hp') or {}
    want = {'model': 'xception', 'tile_px': 299, 'hidden_layers': 2, 'hidden_layer_width': 1024}
    bad = {k: hp[k] for k, v in want.items() if k in hp and hp[k] != v}
    if bad:
        raise ImportError_(f'{p}: hyper-parameters {bad} differ from hp.nature2022 {want}')
    return {'norm_fit': raw.get('norm_fit'), 'normalizer': hp.get('normalizer', raw.get('normalizer')),
            'outcome_labels': raw.get('outcome_labels'), 'outcomes': raw.get('outcomes'), 'hp': hp,
            'tile_px': raw.get('tile_px', hp.get('tile_px')), 'dropout': hp.get('dropout'), 'path': p}


def load_weights(path, verify=True):
    """One entry point for ``--weights``: ``.npz`` (``weights.save_npz`` or Keras-named), ``.safetensors``,
    a checkpoint prefix, a ``variables`` directory or a SavedModel / Slideflow model directory."""
    if os.path.isfile(path) and path.endswith('.npz'):
        with np.load(path) as z:
            return from_named({k: z[k] for k in z.files})
    if os.path.isfile(path) and path.endswith('.safetensors'):
        return load_safetensors(path)
    return from_bundle(path, verify=verify)


def load_model_dir(model_dir, verify=True):
    """(weights, params) of a Slideflow model directory; params is None without a params.json."""
    return from_bundle(model_dir, verify=verify), read_params(model_dir)


def export_bundle(prefix, w, nested=True, optimizer_slots=False, metadata=None, order=None):
    """Write the canonical dict as an object-based Keras checkpoint in Keras' layer order (the inverse of
    ``from_bundle``; ``nested`` puts the backbone under ``layer_with_weights-0`` like a base model called
    inside the classifier).  ``metadata``: also write the object graph and this ``keras_metadata.pb`` path with the Keras
    layer names; ``order``: the weighted layers in another numbering (a permutation of ``keras_layer_order()``).
    Test fixture generator and an exit path back to TensorFlow."""
    tensors = {'save_counter' + _SUFFIX: np.asarray(1, np.int64)}
    idx_backbone = idx_top = 0
    res_block = {name: n for n, (name, _, _) in enumerate(W.residual_plan())}
    nodes = [{'children': {}, 'attributes': {}}]               # object graph: root, [backbone model], layers
    records = [{'node_id': 0, 'node_path': 'root', 'identifier': '_tf_keras_network', 'metadata': {'name': 'model'}}]
    if nested:
        nodes.append({'children': {}, 'attributes': {}})
        nodes[0]['children']['layer_with_weights-0'] = 1
        nodes[0]['children']['layer-1'] = 1
        records.append({'node_id': 1, 'node_path': 'root.layer-1', 'identifier': '_tf_keras_network', 'metadata': {'name': 'xception'}})
    for name, kind in (order or keras_layer_order()):
        top = kind == 'dense' or not nested
        if top:
            n = (idx_top + 1) if nested else idx_top
            path, parent, ppath = f'layer_with_weights-{n}', 0, 'root'
            idx_top += 1
        else:
            n = idx_backbone
            path, parent, ppath = f'layer_with_weights-0/layer_with_weights-{n}', 1, 'root.layer-1'
            idx_backbone += 1
        nid = len(nodes)
        nodes.append({'children': {}, 'attributes': {}})
        nodes[parent]['children'][f'layer_with_weights-{n}'] = nid
        nodes[parent]['children'][f'layer-{2 * n + 2}'] = nid          # some layers without weights in between
        base = name[:-len('_conv')] if name.endswith('_res_conv') else name[:-len('_bn')] if name.endswith('_res_bn') else None
        keras_name = name
        if base in res_block:                                    # Keras' automatic names, offset as after earlier models
            keras_name = ('conv2d' if name.endswith('_conv') else 'batch_normalization') + f'_{res_block[base] + 5}'
        records.append({'node_id': nid, 'node_path': f'{ppath}.layer-{2 * n + 2}', 'identifier': '_tf_keras_layer',
                        'metadata': {'name': keras_name, 'class_name': kind}})
        for k, v in w.items():
            layer, _, var = k.partition('/')
            if layer == name:
                tensors[f'{path}/{var}{_SUFFIX}'] = np.asarray(v, np.float32)
                nodes[nid]['attributes'][var] = f'{path}/{var}{_SUFFIX}'
                if optimizer_slots and var in ('kernel', 'pointwise_kernel'):
                    tensors[f'{path}/{var}/.OPTIMIZER_SLOT/optimizer/m{_SUFFIX}'] = np.zeros_like(v, np.float32)
    if optimizer_slots:
        tensors['optimizer/iter' + _SUFFIX] = np.asarray(7, np.int64)
    tensors['_CHECKPOINTABLE_OBJECT_GRAPH'] = tf_bundle.build_object_graph(nodes) if metadata else b''
    tf_bundle.write_bundle(prefix, tensors)
    if metadata:
        with open(metadata, 'wb') as f:
            f.write(tf_bundle.build_saved_metadata(records))
