"""TEST INFRASTRUCTURE -- writes tests/golden/stain_reinhard.npz: for two smooth synthetic tiles (regenerated
from a seed) and target statistics fitted to a third, oracle/stain.py's CIE-LAB statistics, a SHA-256 of
its uint8 output and a 32x32 crop of it.
The fixture pins the oracle's own arithmetic against drift; it is NOT a reference golden (Slideflow's
normaliser cannot run here: parity unpinned, see oracle/stain.py).  usage: python -m oracle.make_stain_golden
"""
import os

import numpy as np

from oracle import stain


def smooth_tiles(n, seed, px=299):
    """Compressible H&E-like tiles: low-frequency colour fields plus a little noise."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:px, 0:px].astype(np.float32) / px
    out = []
    for i in range(n):
        f = rng.uniform(1.0, 4.0, 6)
        base = np.stack([0.75 + 0.2 * np.sin(f[0] * x * 6 + i) * np.cos(f[1] * y * 5),
                         0.45 + 0.25 * np.cos(f[2] * x * 4) * np.sin(f[3] * y * 7 + i),
                         0.70 + 0.2 * np.sin(f[4] * (x + y) * 5) + 0.05 * np.cos(f[5] * y * 9)], -1)
        noise = rng.integers(-6, 7, (px, px, 3))
        out.append(np.clip(base * 255 + noise, 0, 255).astype(np.uint8))
    return np.stack(out)


def main():
    tiles = smooth_tiles(3, seed=42)
    tm, ts = stain.fit(tiles[2])
    out = stain.reinhard_fast(tiles[:2], tm, ts)
    L, a, b = stain.rgb_to_lab(tiles[:2])
    mu, sd = stain.lab_stats(L, a, b)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden',
                        'stain_reinhard.npz')
    import hashlib
    # the tiles are regenerated from the seed (smooth_tiles above); the fixture keeps what pins the result:
    # target and tile statistics, a digest of the full output and a 32x32 crop of it
    np.savez_compressed(path, seed=np.int64(42), target_means=tm, target_stds=ts, lab_means=mu, lab_stds=sd,
                        input_sha256=np.frombuffer(hashlib.sha256(tiles[:2].tobytes()).digest(), np.uint8),
                        output_sha256=np.frombuffer(hashlib.sha256(out.tobytes()).digest(), np.uint8),
                        output_crop=out[:, 100:132, 100:132])
    print(path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
