# GPU box: five ragged PNG TFRecord slides through the CLI with the host decoder and with --gpu-decode 16; the tile and slide tables
# must be identical (bash tools/cli_gpu_decode_check.sh; prints the row counts and two `True`s).
set -e
cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, numpy as np
from biscuit_amd import tfrecord
from biscuit_amd.synthetic import make_slides
os.makedirs('/tmp/cli/tfr', exist_ok=True)
tiles, sidx, y = make_slides(5, 300, seed=3)
with open('/tmp/cli/labels.csv','w') as f:
    f.write('slide,label\n')
    for i in range(5):
        tfrecord.write_slide(f'/tmp/cli/tfr/s{i}.tfrecords', f's{i}', tiles[sidx==i][: 300 - 37*i])
        f.write(f's{i},{int(y[i])}\n')
PY
python -m biscuit_amd --tfrecords /tmp/cli/tfr --labels /tmp/cli/labels.csv --out /tmp/cli/a --mc 30 2>&1 | grep -v amdgpu | tail -2
python -m biscuit_amd --tfrecords /tmp/cli/tfr --labels /tmp/cli/labels.csv --out /tmp/cli/b --mc 30 --gpu-decode 16 2>&1 | grep -v amdgpu | tail -2
python - <<'PY'
import pandas as pd, glob
a=pd.read_csv(glob.glob('/tmp/cli/a/tile_predictions*.csv')[0]); b=pd.read_csv(glob.glob('/tmp/cli/b/tile_predictions*.csv')[0])
print(len(a), len(b), a.equals(b))
sa=pd.read_csv(glob.glob('/tmp/cli/a/slide_predictions*.csv')[0]); sb=pd.read_csv(glob.glob('/tmp/cli/b/slide_predictions*.csv')[0])
print(sa.equals(sb)); print(sa.head())
PY
