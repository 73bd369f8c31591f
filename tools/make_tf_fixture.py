"""Write the TensorFlow-side pin this repository cannot produce itself (no TensorFlow in the build container or on the
GPU box): a tiny Keras Xception + Slideflow-style UQ head checkpoint and its outputs, as a fixture that

  * ``biscuit_amd.keras_import`` must read (a TF-WRITTEN SavedModel through the importer), and
  * ``oracle/xception_ref.py`` must reproduce (row c: the producer oracle pinned to TensorFlow, not to a restatement).

Run ONCE wherever TensorFlow >= 2.7 is importable (requirements.txt:5), from the repository root:

    python tools/make_tf_fixture.py [out_dir=tests/golden/tf_xception]

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


if __name__ == '__main__':
    main()
