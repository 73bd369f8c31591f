"""The two parsers of untrusted bytes in libbiscuit_io.so -- the inflate of the PNG decoder and the baseline-JPEG decoder --
under AddressSanitizer + UndefinedBehaviorSanitizer on the CPU: tools/fuzz/*.cpp mutate valid streams and decode them from
exact-size heap buffers.  A short run per test invocation (the long runs are quoted in DESIGN.md); skipped where the
compiler has no sanitizer runtime."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ['-O1', '-g', '-std=c++17', '-fsanitize=address,undefined', '-fno-sanitize-recover=undefined',
         '-D_GLIBCXX_SANITIZE_VECTOR']        # poisons size()..capacity() of std::vector: over-reads of a reused buffer show


def _build(src, out, libs=()):
    p = subprocess.run(['g++'] + FLAGS + [os.path.join(ROOT, 'tools', 'fuzz', src), '-o', out] + list(libs),
                       capture_output=True, text=True, timeout=600)
    if p.returncode != 0 and ('asan' in p.stderr.lower() or 'ubsan' in p.stderr.lower() or 'sanitize' in p.stderr.lower()):
        pytest.skip('no sanitizer runtime for g++ here')
    assert p.returncode == 0, p.stderr[-2000:]


def test_inflate_under_sanitizers(tmp_path):
    exe = str(tmp_path / 'inflate_fuzz')
    _build('inflate_fuzz.cpp', exe, ['-lz'])
    p = subprocess.run([exe, '4000'], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-500:], p.stderr[-2000:])
    assert p.stdout.strip().endswith('disagreements 0'), p.stdout[-500:]      # same verdict and bytes as zlib on every stream


def test_jpeg_decoder_under_sanitizers(tmp_path):
    pytest.importorskip('PIL')
    corpus = str(tmp_path / 'corpus')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'fuzz', 'make_jpeg_corpus.py'), corpus], capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-1000:]
    files = sorted(os.path.join(corpus, f) for f in os.listdir(corpus))
    assert len(files) >= 10
    exe = str(tmp_path / 'jpeg_fuzz')
    _build('jpeg_fuzz.cpp', exe)
    p = subprocess.run([exe, '6000'] + files, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-500:], p.stderr[-2000:])        # a sanitizer report aborts with a non-zero status
    assert 'decoded' in p.stdout and 'refused' in p.stdout


def test_table_writer_under_sanitizers(tmp_path):
    """csrc/table_writer.cpp (the output side of libbiscuit_io): the float formatter on 200 000 random doubles / widened float32
    values (each reads back to the same double), tables with awkward slide names, refusals -- no sanitizer report."""
    exe = str(tmp_path / 'table_fuzz')
    _build('table_fuzz.cpp', exe)
    p = subprocess.run([exe, '200000'], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-500:], p.stderr[-2000:])
    assert p.stdout.strip().endswith('problems 0'), p.stdout[-500:]
