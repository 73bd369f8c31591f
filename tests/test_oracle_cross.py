"""The two independent CPU restatements of the producer (``oracle/xception_ref.py``: functional ops, folded
BatchNorm, nested loops; ``oracle/xception_nn.py``: stock torch.nn modules driven by the Keras layer-name
list, unfolded BatchNorm, padding from TensorFlow's formula) must agree layer by layer and end to end, on
the default weights and on the stress set (``synthetic_weights(hard=True)``).  This does not pin the
producer to Slideflow/TensorFlow (impossible here: PARITY UNPINNED), it rules out a single-hand mistake."""
import numpy as np
import pytest
import torch

from biscuit_amd.synthetic import make_tiles
from biscuit_amd.weights import synthetic_weights
from oracle import xception_nn as NN
from oracle.xception_ref import XceptionOracle, standardize

# first oracle's tap name -> Keras layer whose output it is
SAME = {'block1_conv1': 'block1_conv1_act', 'block1_conv2': 'block1_conv2_act',
        'block2_res': 'batch_normalization', 'block3_res': 'batch_normalization_1', 'block4_res': 'batch_normalization_2',
        'block2_sepconv2': 'block2_sepconv2_bn', 'block3_sepconv2': 'block3_sepconv2_bn', 'block4_sepconv2': 'block4_sepconv2_bn',
        'block2_out': 'add', 'block3_out': 'add_1', 'block4_out': 'add_2', 'block13_out': 'add_11',
        'block14_sepconv1': 'block14_sepconv1_act', 'block14_sepconv2': 'block14_sepconv2_act'}
SAME.update({f'block{b}_out': f'add_{b - 2}' for b in range(5, 13)})


@pytest.mark.parametrize('hard', [False, True])
def test_two_oracles_agree(hard):
    w = synthetic_weights(1, hard=hard)
    tiles = make_tiles(2, seed=5)
    ref = XceptionOracle(w)
    net = NN.XceptionNN(w)
    x1, x2 = standardize(tiles), NN.per_image_standardization(tiles)
    assert float((x1 - x2).abs().max()) < 2e-6
    taps, keep = {}, {}
    f1 = ref.backbone(x1, taps)
    f2 = net.features(x2, keep)
    for tap, layer in SAME.items():
        d = float((taps[tap] - keep[layer]).abs().max())
        assert d < 2e-5 * max(1.0, float(keep[layer].abs().max())), (tap, layer, d)
    assert float((f1 - f2).abs().max()) < 5e-6 * max(1.0, float(f1.abs().max()))
    m1, s1 = ref.mc_predict(tiles, 6, 1234, tile_index0=7)
    m2, s2 = NN.mc_predict(net, tiles, 6, 1234, tile_index0=7)
    assert np.abs(m1 - m2).max() < 1e-6 and np.abs(s1 - s2).max() < 1e-6
    if hard:        # the stress set really stresses: predictions far from 0.5, MC std of order 0.1
        assert np.abs(m1[:, 1] - 0.5).max() > 0.05 and s1.max() > 0.05


def test_layer_list_is_keras_xception():
    L = NN.keras_xception_layers()
    names = [n for n, *_ in L]
    assert len(names) == len(set(names)) == 132             # + input_1 = the 133 layers Keras reports with pooling='avg'
    assert names[:3] == ['block1_conv1', 'block1_conv1_bn', 'block1_conv1_act'] and names[-1] == 'avg_pool'
    assert sum(k == 'sep' for _, k, *_ in L) == 34 and sum(k == 'add' for _, k, *_ in L) == 12
    assert [n for n in names if n.startswith('conv2d')] == ['conv2d', 'conv2d_1', 'conv2d_2', 'conv2d_3']
    # Keras lists a block's residual convolution BEFORE the block's separable convolutions (model.layers order)
    assert names.index('conv2d') < names.index('block2_sepconv1') < names.index('add')
    net = NN.XceptionNN(synthetic_weights(1))
    n_params = sum(p.numel() for n, p in net.named_parameters() if n.startswith('mods.')) + \
        sum(b.numel() for n, b in net.named_buffers() if n.endswith(('running_mean', 'running_var')))
    assert n_params == 20861480                             # keras.applications.Xception(include_top=False)


def test_same_padding_rule():
    # TensorFlow 'same': asymmetric where the total is odd (the extra row/column goes after)
    assert NN._same_pad(147, 3, 2) == (1, 1) and NN._same_pad(74, 3, 2) == (0, 1)
    assert NN._same_pad(37, 3, 2) == (1, 1) and NN._same_pad(19, 3, 2) == (1, 1)
    assert NN._same_pad(147, 1, 2) == (0, 0) and NN._same_pad(74, 1, 2) == (0, 0)
    assert NN._same_pad(19, 3, 1) == (1, 1)
