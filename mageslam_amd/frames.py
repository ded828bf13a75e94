"""Deterministic synthetic camera frames for the ORB / matching configuration.

A frame is a smooth gradient plus 200 random axis-aligned rectangles and 300 discs of random grey level, plus
N(0, 2^2) pixel noise, clamped to uint8.  ``warp`` resamples a frame through a small homography (bilinear) so a
second view with true correspondences exists.  Everything is drawn from the same counter-based SplitMix64
streams as scene.py, so a (seed, frame index) pair always yields the same bytes.
"""
from __future__ import annotations

import numpy as np

from .scene import normal, uniform

_S_RECT, _S_DISC, _S_PIX = 11, 12, 13


def make_frame(seed: int, width: int = 640, height: int = 480, n_rect: int = 200, n_disc: int = 300, noise: float = 2.0) -> np.ndarray:
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float64)
    img = 60.0 + 90.0 * xx / max(width - 1, 1) + 50.0 * yy / max(height - 1, 1)
    i = np.arange(n_rect)
    x0 = (uniform(seed, _S_RECT, i, 0) * width).astype(int); y0 = (uniform(seed, _S_RECT, i, 1) * height).astype(int)
    rw = (4 + uniform(seed, _S_RECT, i, 2) * width * 0.15).astype(int); rh = (4 + uniform(seed, _S_RECT, i, 3) * height * 0.15).astype(int)
    g = uniform(seed, _S_RECT, i, 4) * 255.0
    for k in range(n_rect):
        img[y0[k]: y0[k] + rh[k], x0[k]: x0[k] + rw[k]] = g[k]
    i = np.arange(n_disc)
    cx = uniform(seed, _S_DISC, i, 0) * width; cy = uniform(seed, _S_DISC, i, 1) * height
    rad = 2.0 + uniform(seed, _S_DISC, i, 2) * 0.05 * min(width, height)
    g = uniform(seed, _S_DISC, i, 3) * 255.0
    for k in range(n_disc):
        xa, xb = max(int(cx[k] - rad[k]) - 1, 0), min(int(cx[k] + rad[k]) + 2, width)
        ya, yb = max(int(cy[k] - rad[k]) - 1, 0), min(int(cy[k] + rad[k]) + 2, height)
        if xa >= xb or ya >= yb:
            continue
        m = (xx[ya:yb, xa:xb] - cx[k]) ** 2 + (yy[ya:yb, xa:xb] - cy[k]) ** 2 <= rad[k] ** 2
        img[ya:yb, xa:xb][m] = g[k]
    p = np.arange(width * height)
    img = img + noise * normal(seed, _S_PIX, p, 0).reshape(height, width)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def warp(img: np.ndarray, H: np.ndarray) -> np.ndarray:
    """dst(x, y) = bilinear sample of img at H^-1 (x, y, 1); outside pixels take the border value."""
    h, w = img.shape
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    Hi = np.linalg.inv(H)
    X = Hi[0, 0] * xx + Hi[0, 1] * yy + Hi[0, 2]
    Y = Hi[1, 0] * xx + Hi[1, 1] * yy + Hi[1, 2]
    Z = Hi[2, 0] * xx + Hi[2, 1] * yy + Hi[2, 2]
    X = np.clip(X / Z, 0, w - 1.001); Y = np.clip(Y / Z, 0, h - 1.001)
    x0 = np.floor(X).astype(int); y0 = np.floor(Y).astype(int)
    fx, fy = X - x0, Y - y0
    f = img.astype(np.float64)
    out = (f[y0, x0] * (1 - fx) * (1 - fy) + f[y0, x0 + 1] * fx * (1 - fy) + f[y0 + 1, x0] * (1 - fx) * fy + f[y0 + 1, x0 + 1] * fx * fy)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def small_homography(seed: int, width: int = 640, height: int = 480) -> np.ndarray:
    """A few pixels of translation, ~1 degree of rotation, slight perspective."""
    u = uniform(seed, 14, np.arange(6), 0) - 0.5
    a = 0.03 * u[0]
    H = np.array([[np.cos(a), -np.sin(a), 6.0 * u[1]], [np.sin(a), np.cos(a), 6.0 * u[2]], [2e-5 * u[3], 2e-5 * u[4], 1.0]])
    C = np.array([[1, 0, width / 2], [0, 1, height / 2], [0, 0, 1.0]])
    return C @ H @ np.linalg.inv(C)


def frame_pair(seed: int, width: int = 640, height: int = 480):
    a = make_frame(seed, width, height)
    b = warp(a, small_homography(seed, width, height))
    # fresh sensor noise on the second view
    p = np.arange(width * height)
    b = np.clip(b.astype(np.float64) + 2.0 * normal(seed + 1, _S_PIX, p, 1).reshape(height, width), 0, 255)
    return a, np.rint(b).astype(np.uint8)
