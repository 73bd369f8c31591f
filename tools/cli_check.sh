#!/bin/bash
# end-to-end CLI run on self-written TFRecords (native reader, pinned prefetch, stain normaliser, pool)
set -e
D=$(mktemp -d)
python - "$D" <<'PY'
import sys, json, numpy as np
from biscuit_amd import tfrecord as tfr
from biscuit_amd.synthetic import make_slides
d = sys.argv[1]
tiles, sidx, y = make_slides(4, 9, seed=3)
rows = []
for i in range(4):
    t = tiles[sidx == i]
    tfr.write_slide(f'{d}/s{i}.tfrecords', f's{i}', t, np.arange(2 * len(t)).reshape(-1, 2))
    rows.append(f's{i},{int(y[i])},p{i // 2}')
open(f'{d}/labels.csv', 'w').write('slide,label,patient\n' + '\n'.join(rows) + '\n')
json.dump({'norm_fit': {'target_means': [65.0, 12.0, -8.0], 'target_stds': [14.0, 7.0, 6.0]}}, open(f'{d}/params.json', 'w'))
PY
python -m biscuit_amd --tfrecords $D --labels $D/labels.csv --out $D/eval --mc 4 --batch 8 --params $D/params.json 2>&1 | grep -v amdgpu | tail -2
python -m biscuit_amd --tfrecords $D --labels $D/labels.csv --out $D/eval2 --mc 4 --batch 8 --streams 1 2>&1 | grep -v amdgpu | tail -1
head -3 $D/eval/tile_predictions_eval.csv; wc -l $D/eval/tile_predictions_eval.csv; cat $D/eval/slide_predictions_cohort_eval.csv | head -5
# the same slides through --model: a Keras-ordered checkpoint of the default synthetic weights + params.json
python - "$D" <<'PY'
import sys, json, os
from biscuit_amd import keras_import as K, weights as W
d = sys.argv[1] + '/00001-cohort-HP0/cohort-HP0_epoch1'
K.export_bundle(d + '/variables/variables', W.synthetic_weights(1), optimizer_slots=True)
json.dump({'norm_fit': {'target_means': [65.0, 12.0, -8.0], 'target_stds': [14.0, 7.0, 6.0]},
           'hp': {'model': 'xception', 'tile_px': 299, 'hidden_layers': 2, 'hidden_layer_width': 1024, 'dropout': 0.1, 'normalizer': 'reinhard_fast'}},
          open(d + '/params.json', 'w'))
PY
python -m biscuit_amd --tfrecords $D --labels $D/labels.csv --out $D/eval3 --mc 4 --batch 8 --model $D/00001-cohort-HP0/cohort-HP0_epoch1 2>&1 | grep -v amdgpu | tail -1
cmp $D/eval/tile_predictions_eval.csv $D/eval3/tile_predictions_eval.csv && echo "--model == --params + default weights: identical tile table"
