"""The TensorFlow pin (SURVEY.md section 8 rows c and f2).  Needs the fixture ``tools/make_tf_fixture.py`` writes where
TensorFlow is installed (``tests/golden/tf_xception`` or ``$BQ_TF_FIXTURE``); skipped while it does not exist -- there is
no TensorFlow in the build container or on the GPU box, and nothing here imports it.

With the fixture present these tests turn "parity unpinned" into a measured figure:
  * the importer reads a TF-WRITTEN SavedModel (layout, object graph, keras_metadata) and binds every layer;
  * the CPU oracle reproduces TensorFlow's standardisation, Xception features (every tapped block) and dropout-free
    probabilities from those weights to fp32 accuracy;
  * (``-m gpu``) the HIP path reproduces the same numbers within the north-star tolerance.
"""
import os

import numpy as np
import pytest

FIX = os.environ.get('BQ_TF_FIXTURE') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'tf_xception')
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(FIX, 'io.npz')),
                                reason='TensorFlow fixture absent: run tools/make_tf_fixture.py where TensorFlow is installed')


@pytest.fixture(scope='module')
def fixture():
    from biscuit_amd import keras_import as K
    return K.from_bundle(os.path.join(FIX, 'saved_model')), np.load(os.path.join(FIX, 'io.npz'))


def test_importer_reads_a_tensorflow_written_savedmodel(fixture):
    from biscuit_amd import weights as W
    w, _ = fixture
    assert W.count_backbone_params(w) == 20_861_480                        # Keras' own count for Xception without top
    assert w['hidden_0/kernel'].shape == (2048, 1024) and w['logits/kernel'].shape == (1024, 2)
    W.pack_blob(w, 'f16')                                                    # every tensor the device needs is there


def test_oracle_reproduces_tensorflow(fixture):
    import torch
    from oracle.xception_ref import XceptionOracle, standardize
    w, io = fixture
    x = standardize(io['tiles'])
    np.testing.assert_allclose(x.permute(0, 2, 3, 1).numpy(), io['standardized'], atol=2e-6)
    orc = XceptionOracle(w, dropout=0.0)
    taps = {}
    feat = orc.backbone(x, taps)
    for name in ('block1_conv1', 'block1_conv2', 'block2_out', 'block3_out', 'block4_out', 'block5_out', 'block12_out',
                 'block13_out', 'block14_sepconv2'):
        ref = io[f'tap_{name}']
        got = taps[name].permute(0, 2, 3, 1).numpy()
        if name in ('block1_conv1', 'block1_conv2', 'block14_sepconv2'):   # Keras taps the activation layer's output
            got = np.maximum(got, 0)
        assert np.abs(got - ref).max() < 2e-4 * max(1.0, np.abs(ref).max()), name
    assert np.abs(feat.numpy() - io['features']).max() < 2e-4 * max(1.0, np.abs(io['features']).max())
    h = feat
    for name in ('hidden_0', 'hidden_1'):
        h = torch.relu(h @ torch.from_numpy(w[name + '/kernel']) + torch.from_numpy(w[name + '/bias']))
    p = torch.softmax(h @ torch.from_numpy(w['logits/kernel']) + torch.from_numpy(w['logits/bias']), 1).numpy()
    assert np.abs(p - io['probs_nodrop']).max() < 1e-5


@pytest.mark.gpu
def test_hip_path_reproduces_tensorflow(fixture):
    import torch
    from biscuit_amd.engine import Engine
    from biscuit_amd.hp import ModelParams
    w, io = fixture
    d = torch.from_numpy(io['tiles']).cuda()
    for dtype, tol in (('f32', 2e-4), ('f16', 3e-3)):
        eng = Engine(w, hp=ModelParams(dropout=0.0), dtype=dtype, max_batch=8, max_mc=2)
        feat = eng.backbone(eng.stage(d)).cpu().numpy()
        assert np.abs(feat - io['features']).max() < tol * max(1.0, np.abs(io['features']).max()), dtype
        m, s = eng.mc_infer(d, 2, 1)                                         # dropout 0: both passes = the plain forward
        assert np.abs(m.cpu().numpy() - io['probs_nodrop']).max() < (1e-5 if dtype == 'f32' else 1e-3)
        assert float(s.abs().max()) == 0.0
        eng.close()
