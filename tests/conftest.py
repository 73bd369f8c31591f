import gzip
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
if os.path.join(ROOT, 'tests') not in sys.path:        # helper modules next to the tests (_rank_worker, test_gpu_parity.TAPS)
    sys.path.insert(1, os.path.join(ROOT, 'tests'))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    with gzip.open(os.path.join(GOLDEN, name), 'rt') as f:
        return json.load(f)


@pytest.fixture(scope='session')
def consumer_cases():
    return load_golden('consumer_cases.json.gz')


@pytest.fixture(scope='session')
def consumer_cv():
    return load_golden('consumer_cv.json.gz')
