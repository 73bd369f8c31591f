"""TEST INFRASTRUCTURE -- CPU restatement of the `reinhard_fast` stain normaliser.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

What it restates: hp.py:19 selects ``normalizer='reinhard_fast'`` and results.py:251-256 applies
``interface.wsi_normalizer.rgb_to_rgb(image)`` to the uint8 tile *before*
``tf.image.per_image_standardization``.  The normaliser itself lives in Slideflow
(``slideflow.norm`` -- ``requirements.txt:1``, ``slideflow>=1.1.0rc1``, no lockfile, not vendored, not
installable here), so this is its published algorithm restated from the method's definition
(Reinhard et al. 2001 colour transfer in CIE-LAB, "fast" = without the brightness-standardisation
pre-step):

    lab      = rgb_to_lab(tile / 255)                         (sRGB, D65, 2-degree observer)
    mu, sd   = per-channel mean and population std of lab over the tile
    lab'     = (lab - mu) * (target_std / sd) + target_mean
    out      = clip(int(lab_to_rgb(lab') * 255), 0, 255)      (float -> int truncates, as tf.cast does)

**Parity unpinned**: the reference holds no golden vectors for this step and Slideflow cannot be run
here; ``target_means`` / ``target_stds`` come from a model's ``params.json`` ``norm_fit`` and must be
read from there, never hard-coded.

Precision contract shared with the HIP kernel (so the uint8 results can be compared exactly):
  * sRGB -> linear through a 256-entry table evaluated in float64 and rounded to float32;
  * cube roots and the 1/2.4 power evaluated in float64 and rounded to float32;
  * channel statistics accumulated in float64, rounded to float32;
  * everything else float32, one rounding per operation, in the order written below.
"""
import numpy as np

F = np.float32

# sRGB -> XYZ (D65), the matrix used by scikit-image / tensorflow-io colour code
XYZ_FROM_RGB = np.array([[0.412453, 0.357580, 0.180423],
                         [0.212671, 0.715160, 0.072169],
                         [0.019334, 0.119193, 0.950227]], dtype=np.float64)
RGB_FROM_XYZ = np.linalg.inv(XYZ_FROM_RGB)
WHITE_D65 = np.array([0.95047, 1.0, 1.08883], dtype=np.float64)


def srgb_to_linear_lut():
    c = np.arange(256, dtype=np.float64) / 255.0
    lin = np.where(c > 0.04045, ((c + 0.055) / 1.055) ** 2.4, c / 12.92)
    return lin.astype(F)


def constants():
    """The float32 constants both implementations use (the C side derives the same values)."""
    return {'lut': srgb_to_linear_lut(), 'm': XYZ_FROM_RGB.astype(F), 'minv': RGB_FROM_XYZ.astype(F),
            'white': WHITE_D65.astype(F)}


def _cbrt32(x):
    return np.cbrt(x.astype(np.float64)).astype(F)


def rgb_to_lab(tiles_u8):
    """uint8 [..., 3] -> float32 L, a, b arrays."""
    k = constants()
    lin = k['lut'][tiles_u8]                                       # [..., 3] float32
    r, g, b = lin[..., 0], lin[..., 1], lin[..., 2]
    m = k['m']
    xyz = [F(m[i, 0]) * r + F(m[i, 1]) * g + F(m[i, 2]) * b for i in range(3)]   # (r*m0 + g*m1) + b*m2
    f = []
    for i in range(3):
        t = xyz[i] / k['white'][i]
        f.append(np.where(t > F(0.008856), _cbrt32(t), F(7.787) * t + F(16.0 / 116.0)).astype(F))
    L = F(116.0) * f[1] - F(16.0)
    a = F(500.0) * (f[0] - f[1])
    bb = F(200.0) * (f[1] - f[2])
    return L.astype(F), a.astype(F), bb.astype(F)


def lab_stats(L, a, b):
    """Per-tile channel means and population stds: arrays [n, 3] float32 (float64 accumulation)."""
    means, stds = [], []
    for ch in (L, a, b):
        c = ch.reshape(ch.shape[0], -1).astype(np.float64)
        mu = c.mean(axis=1)
        var = np.maximum((c * c).mean(axis=1) - mu * mu, 0.0)
        means.append(mu.astype(F))
        stds.append(np.sqrt(var).astype(F))
    return np.stack(means, 1), np.stack(stds, 1)


def lab_to_rgb_u8(L, a, b):
    k = constants()
    fy = (L + F(16.0)) / F(116.0)
    fx = a / F(500.0) + fy
    fz = fy - b / F(200.0)
    xyz = []
    for i, v in enumerate((fx, fy, fz)):
        t = np.where(v > F(0.2068966), (v * v) * v, (v - F(16.0 / 116.0)) / F(7.787)).astype(F)
        xyz.append(t * k['white'][i])
    mi = k['minv']
    out = []
    for i in range(3):
        c = (F(mi[i, 0]) * xyz[0] + F(mi[i, 1]) * xyz[1] + F(mi[i, 2]) * xyz[2]).astype(F)
        big = c > F(0.0031308)
        p = np.power(np.where(big, c, F(1.0)).astype(np.float64), 1.0 / 2.4).astype(F)
        c = np.where(big, F(1.055) * p - F(0.055), c * F(12.92)).astype(F)
        c = np.clip(c, F(0.0), F(1.0))
        v = (c * F(255.0)).astype(F)
        out.append(np.clip(np.trunc(v), 0, 255).astype(np.uint8))
    return np.stack(out, -1)


def fit(target_u8):
    """Target statistics of one image [H, W, 3] uint8 -> (target_means[3], target_stds[3])."""
    L, a, b = rgb_to_lab(target_u8[None])
    mu, sd = lab_stats(L, a, b)
    return mu[0], sd[0]


def reinhard_fast(tiles_u8, target_means, target_stds):
    """uint8 [n, H, W, 3] -> uint8 [n, H, W, 3]."""
    tiles_u8 = np.asarray(tiles_u8, dtype=np.uint8)
    tm = np.asarray(target_means, F)
    ts = np.asarray(target_stds, F)
    L, a, b = rgb_to_lab(tiles_u8)
    mu, sd = lab_stats(L, a, b)
    chans = []
    for i, ch in enumerate((L, a, b)):
        scale = (ts[i] / sd[:, i]).astype(F)[:, None, None]
        chans.append(((ch - mu[:, i][:, None, None]) * scale + tm[i]).astype(F))
    return lab_to_rgb_u8(*chans)
