"""numpy restatement of the reference's CPU preprocessing (TEST INFRASTRUCTURE).

Functions follow mmfn_utils/datasets/dataloader.py (file:line cited per function).
They are the checkers for the HIP ingest/splat kernels; nothing here is shipped.
"""
import numpy as np


def lidar_histogram(points, bins=256):
    """dataloader.py:271-293 (lidar_to_histogram_features).

    points [N,>=3] -> f32 [2, 256, 256] (channel 0: z <= -2, channel 1: z > -2; axis 1 = x
    bin over [-16,16], axis 2 = y bin over [-24,8]; half-open bins, last bin right-closed as
    np.histogramdd does; counts clipped at 5 and divided by 5).
    """
    points = np.asarray(points)
    xe = np.linspace(-16.0, 16.0, bins + 1)
    ye = np.linspace(-24.0, 8.0, bins + 1)
    out = np.zeros((2, bins, bins), dtype=np.float64)
    for ch, sel in enumerate((points[..., 2] <= -2.0, points[..., 2] > -2.0)):
        p = points[sel]
        ix = np.searchsorted(xe, p[:, 0], side="right") - 1
        iy = np.searchsorted(ye, p[:, 1], side="right") - 1
        ix[p[:, 0] == xe[-1]] = bins - 1
        iy[p[:, 1] == ye[-1]] = bins - 1
        ok = (ix >= 0) & (ix < bins) & (iy >= 0) & (iy < bins)
        np.add.at(out[ch], (ix[ok], iy[ok]), 1.0)
    return (np.minimum(out, 5.0) / 5.0).astype(np.float32)


def crop_chw(image_hwc, crop=256):
    """dataloader.py:296-308 (scale_and_crop_image, scale=1): centre crop, HWC -> CHW."""
    h, w = image_hwc.shape[:2]
    r0, c0 = h // 2 - crop // 2, w // 2 - crop // 2
    return np.transpose(image_hwc[r0:r0 + crop, c0:c0 + crop], (2, 0, 1))


def radar_to_size(data, rows=81, cols=5):
    """dataloader.py:336-346: pad with zero rows, or drop the rows with largest |c0/c3|."""
    data = np.asarray(data)
    if data.shape[0] >= rows:
        n = data.shape[0] - rows
        return np.delete(data, (-abs(data[:, 0] / data[:, 3])).argsort()[:n], 0)
    out = np.zeros((rows, cols))
    out[:data.shape[0]] = data
    return out


def radar_adjacency(radar):
    """dataloader.py:381-384: adj[i, j] = radar[j, 1] - radar[i, 1]."""
    radar = np.asarray(radar)
    return radar[None, :, 1] - radar[:, None, 1]
