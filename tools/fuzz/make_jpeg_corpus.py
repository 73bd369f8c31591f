"""Seed files for tools/fuzz/jpeg_fuzz.cpp: python tools/fuzz/make_jpeg_corpus.py OUTDIR"""
import io
import os
import sys

import numpy as np
from PIL import Image


def main(out):
    os.makedirs(out, exist_ok=True)
    rng = np.random.default_rng(0)
    n = 0
    for px in (299, 72, 17):
        y, x = np.mgrid[0:px, 0:px]
        photo = np.stack([128 + 100 * np.sin(x / 17.0 + c) + 20 * np.cos(y / 9.0 * c + 1) for c in range(3)], -1)
        photo = np.clip(photo + rng.normal(0, 12, photo.shape), 0, 255).astype(np.uint8)
        noise = rng.integers(0, 256, (px, px, 3), dtype=np.uint8)
        for img in (photo, noise, photo[..., 0]):
            for q, ss, kw in ((85, 2, {}), (95, 0, {'optimize': True}), (40, 1, {'restart_marker_blocks': 3})):
                b = io.BytesIO()
                try:
                    Image.fromarray(img).save(b, format='JPEG', quality=q, **({} if img.ndim == 2 else {'subsampling': ss}), **kw)
                except OSError:
                    continue
                open(os.path.join(out, f'seed{n:02d}_{px}.jpg'), 'wb').write(b.getvalue())
                n += 1
    print(n, 'files in', out)


if __name__ == '__main__':
    main(sys.argv[1])
