"""Hyper-parameters of the classifier on the hot path: the fields of
``biscuit.hp.nature2022`` (``biscuit/hp.py:3-23``) that determine what the kernels
compute, plus the MC-dropout controls Slideflow keeps elsewhere (``uq_n`` = 30 passes).
Training-only fields (optimizer, learning rate, augment, early stopping) are not part
of the inference path and are omitted."""
from dataclasses import dataclass


@dataclass
class ModelParams:
    model: str = 'xception'            # hp.py:4
    tile_px: int = 299                 # hp.py:5
    tile_um: int = 302                 # hp.py:6
    batch_size: int = 128              # hp.py:7
    dropout: float = 0.1               # hp.py:11
    uq: bool = False                   # hp.py:12; set True by experiment.py:849,875,891
    hidden_layers: int = 2             # hp.py:21
    hidden_layer_width: int = 1024     # hp.py:13
    pooling: str = 'avg'               # hp.py:22
    include_top: bool = False          # hp.py:20
    normalizer: str = 'reinhard_fast'  # hp.py:19 (stain normaliser, applied before staging)
    uq_n: int = 30                     # Slideflow's number of MC-dropout passes
    seed: int = 1234                   # Philox key of the dropout masks

    def validate(self):
        if (self.model, self.tile_px, self.hidden_layers, self.hidden_layer_width, self.pooling,
                self.include_top) != ('xception', 299, 2, 1024, 'avg', False):
            raise ValueError('libbiscuit_hip implements exactly the biscuit.hp.nature2022 '
                             'architecture (xception/299px/avg-pool/2x1024 hidden)')
        if not (0.0 <= self.dropout < 1.0):
            raise ValueError('dropout must be in [0, 1)')
        return self


def nature2022():
    """The configuration of ``biscuit.hp.nature2022`` (``biscuit/hp.py:3``)."""
    return ModelParams()
