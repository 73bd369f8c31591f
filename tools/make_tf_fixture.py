"""Write the TensorFlow-side pin this repository cannot produce itself (no TensorFlow in the build container or on the
GPU box): a tiny Keras Xception + Slideflow-style UQ head checkpoint and its outputs, as a fixture that

  * ``biscuit_amd.keras_import`` must read (a TF-WRITTEN SavedModel through the importer), and
  * ``oracle/xception_ref.py`` must reproduce (row c: the producer oracle pinned to TensorFlow, not to a restatement).

Run ONCE wherever TensorFlow >= 2.7 is importable (requirements.txt:5), from the repository root:

    python tools/make_tf_fixture.py [out_dir=tests/golden/tf_xception]
    python tools/make_tf_fixture.py --check [out_dir]     # needs no TensorFlow: keys, shapes, dtypes, consistency, the importer

(the writer runs the check on what it wrote and fails loudly if it does not validate)

and commit the directory (about 95 MB of variables -- or keep it out of git and point BQ_TF_FIXTURE at it).  The tests
``tests/test_tf_fixture.py`` are skipped while the fixture is absent and need no TensorFlow themselves.

What is written (all seeded, float32):
  saved_model/            keras ``model.save`` of Input(299,299,3) -> keras.applications.Xception(include_top=False,
                          pooling='avg', weights=None) -> Dropout(0.1) -> Dense(1024, relu, 'hidden_0') -> Dropout(0.1)
                          -> Dense(1024, relu, 'hidden_1') -> Dropout(0.1) -> Dense(2, 'logits') -> softmax, the
                          architecture biscuit/hp.py:3-23 asks Slideflow for; BatchNorm moving statistics, gammas and
                          betas randomised (a fresh model's are the identity and would pin nothing)
  params.json             hp block + a norm_fit, as Slideflow writes next to a model
  io.npz                  tiles uint8 [4,299,299,3]; standardized = tf.image.per_image_standardization(tiles)
                          (results.py:256); features = the pooled 2048-vector; probs_nodrop = model(x, training=False)
                          with the dropout layers inert; taps: the outputs of a few named Xception layers
The MC loop itself (30 passes, reduce_mean / reduce_std, results.py:257-258 and Slideflow's get_uq_predictions) uses
TensorFlow's own random stream, which nothing outside TensorFlow can reproduce: it is pinned through features and
deterministic probabilities, not through sampled masks.
"""
import json
import os
import sys

import numpy as np

TAPS = ['block1_conv1_act', 'block1_conv2_act', 'add', 'add_1', 'add_2', 'add_3', 'add_10', 'add_11', 'block14_sepconv2_act']
NAMES = ['block1_conv1', 'block1_conv2', 'block2_out', 'block3_out', 'block4_out', 'block5_out', 'block12_out', 'block13_out',
         'block14_sepconv2']


def main():
    import tensorflow as tf
    from tensorflow import keras
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join('tests', 'golden', 'tf_xception')
    os.makedirs(out, exist_ok=True)
    tf.keras.utils.set_random_seed(20221003)
    core = keras.applications.Xception(include_top=False, weights=None, pooling='avg', input_shape=(299, 299, 3))
    rng = np.random.default_rng(7)
    for layer in core.layers:                          # BatchNorm far from the identity, like a trained model's
        if isinstance(layer, keras.layers.BatchNormalization):
            g, b, m, v = layer.get_weights()
            layer.set_weights([rng.uniform(0.5, 1.5, g.shape).astype(np.float32), rng.normal(0, 0.2, b.shape).astype(np.float32),
                               rng.normal(0, 0.3, m.shape).astype(np.float32), np.exp(rng.uniform(-1.4, 1.4, v.shape)).astype(np.float32)])
    inp = keras.Input((299, 299, 3), name='tile_image')
    x = core(inp)
    for i in range(2):
        x = keras.layers.Dropout(0.1, name=f'dropout_{i}')(x)
        x = keras.layers.Dense(1024, activation='relu', name=f'hidden_{i}')(x)
    x = keras.layers.Dropout(0.1, name='dropout_2')(x)
    x = keras.layers.Dense(2, name='logits')(x)
    probs = keras.layers.Activation('softmax', dtype='float32', name='out-0')(x)
    model = keras.Model(inp, probs)
    model.save(os.path.join(out, 'saved_model'))
    json.dump({'hp': {'model': 'xception', 'tile_px': 299, 'tile_um': 302, 'hidden_layers': 2, 'hidden_layer_width': 1024,
                      'dropout': 0.1, 'pooling': 'avg', 'include_top': False, 'normalizer': 'reinhard_fast', 'uq': True},
               'norm_fit': {'target_means': [65.0, 12.0, -8.0], 'target_stds': [14.0, 7.0, 6.0]},
               'outcome_labels': {'0': 'LUAD', '1': 'LUSC'}, 'tensorflow': tf.__version__},
              open(os.path.join(out, 'params.json'), 'w'), indent=1)
    tiles = rng.integers(0, 256, (4, 299, 299, 3), dtype=np.uint8)
    tiles[1] = (tiles[1] * 0.3 + 150).astype(np.uint8)          # a low-contrast tile
    std = tf.image.per_image_standardization(tf.convert_to_tensor(tiles)).numpy().astype(np.float32)
    feat_model = keras.Model(core.input, [core.get_layer(n).output for n in TAPS] + [core.output])
    *taps, feat = feat_model(std, training=False)
    p = model(std, training=False).numpy()
    np.savez_compressed(os.path.join(out, 'io.npz'), tiles=tiles, standardized=std, features=feat.numpy(), probs_nodrop=p,
                        **{f'tap_{n}': t.numpy() for n, t in zip(NAMES, taps)})
    print('wrote', out, '; tensorflow', tf.__version__)


# ---- --check: validate a fixture WITHOUT TensorFlow (so the first box that has TensorFlow cannot leave a silently unusable file)
TAP_SHAPES = {'block1_conv1': (149, 149, 32), 'block1_conv2': (147, 147, 64), 'block2_out': (74, 74, 128),
              'block3_out': (37, 37, 256), 'block4_out': (19, 19, 728), 'block5_out': (19, 19, 728),
              'block12_out': (19, 19, 728), 'block13_out': (10, 10, 1024), 'block14_sepconv2': (10, 10, 2048)}


def check_io(io, n=None):
    """Problems of an ``io.npz`` (a dict-like of arrays): missing keys, wrong shapes / dtypes, values that cannot be what
    the key says.  Returns a list of strings, empty when the file is usable by tests/test_tf_fixture.py."""
    bad = []

    def need(key, shape, dtype):
        if key not in io:
            bad.append(f'missing array {key!r}')
            return None
        a = np.asarray(io[key])
        if tuple(a.shape) != tuple(shape):
            bad.append(f'{key}: shape {tuple(a.shape)}, want {tuple(shape)}')
        if a.dtype != np.dtype(dtype):
            bad.append(f'{key}: dtype {a.dtype}, want {np.dtype(dtype)}')
        if a.dtype.kind == 'f' and not np.isfinite(a).all():
            bad.append(f'{key}: NaN / Inf')
        return a
    if 'tiles' not in io:
        return ["missing array 'tiles'"]
    n = int(np.asarray(io['tiles']).shape[0]) if n is None else n
    tiles = need('tiles', (n, 299, 299, 3), np.uint8)
    std = need('standardized', (n, 299, 299, 3), np.float32)
    need('features', (n, 2048), np.float32)
    p = need('probs_nodrop', (n, 2), np.float32)
    for name in NAMES:
        need(f'tap_{name}', (n,) + TAP_SHAPES[name], np.float32)
    if not bad:
        if n < 2:
            bad.append('fewer than 2 tiles')
        # tf.image.per_image_standardization: zero mean, unit variance (floor 1/sqrt(N)) per tile
        m = std.reshape(n, -1).astype(np.float64)
        if np.abs(m.mean(1)).max() > 1e-3 or np.abs(m.std(1) - 1).max() > 1e-2:
            bad.append('standardized: not zero-mean / unit-variance per tile')
        x = tiles.reshape(n, -1).astype(np.float64)
        ref = (x - x.mean(1, keepdims=True)) / np.maximum(x.std(1, keepdims=True), 1 / np.sqrt(x.shape[1]))
        if np.abs(ref - m).max() > 1e-3:
            bad.append('standardized does not belong to tiles')
        if np.abs(p.sum(1) - 1).max() > 1e-4 or (p < 0).any():
            bad.append('probs_nodrop: rows are not probability vectors')
        if float(np.asarray(io['features']).min()) < 0:
            bad.append('features: negative values behind a ReLU + average pool')
        for name in ('block1_conv1', 'block1_conv2', 'block14_sepconv2'):     # Keras taps the activation's output
            if float(np.asarray(io[f'tap_{name}']).min()) < 0:
                bad.append(f'tap_{name}: negative values in the output of an activation layer')
    return bad


def check(out):
    """Validate the fixture directory ``out``; returns the list of problems (prints them too)."""
    bad = []
    for f in ('io.npz', 'params.json', os.path.join('saved_model', 'saved_model.pb'),
              os.path.join('saved_model', 'variables', 'variables.index')):
        if not os.path.exists(os.path.join(out, f)):
            bad.append(f'missing file {f}')
    if os.path.exists(os.path.join(out, 'io.npz')):
        with np.load(os.path.join(out, 'io.npz')) as z:
            bad += check_io({k: z[k] for k in z.files})
    if os.path.exists(os.path.join(out, 'params.json')):
        try:
            pj = json.load(open(os.path.join(out, 'params.json')))
            if 'norm_fit' not in pj or 'hp' not in pj:
                bad.append('params.json: no hp / norm_fit block')
        except ValueError as e:
            bad.append(f'params.json: {e}')
    if not any(b.startswith('missing file saved_model') for b in bad):
        try:                                              # the importer binds every layer of the TF-written checkpoint
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
            from biscuit_amd import keras_import as K
            from biscuit_amd import weights as W
            w = K.from_bundle(os.path.join(out, 'saved_model'))
            if W.count_backbone_params(w) != 20_861_480:
                bad.append(f'checkpoint: {W.count_backbone_params(w)} backbone parameters, want 20861480')
        except Exception as e:                            # noqa: BLE001 -- report, do not crash the checker
            bad.append(f'checkpoint does not import: {type(e).__name__}: {e}')
    for b in bad:
        print('FIXTURE PROBLEM:', b)
    print('fixture', out, 'is usable' if not bad else f'has {len(bad)} problem(s)')
    return bad


if __name__ == '__main__':
    if '--check' in sys.argv[1:]:
        rest = [a for a in sys.argv[1:] if a != '--check']
        sys.exit(1 if check(rest[0] if rest else os.path.join('tests', 'golden', 'tf_xception')) else 0)
    main()
    if check(sys.argv[1] if len(sys.argv) > 1 else os.path.join('tests', 'golden', 'tf_xception')):
        sys.exit('the fixture just written does not validate: do not commit it')
