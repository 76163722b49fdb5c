"""Builds a smooth natural-like 256x256 test image (gradients, blobs, edges) as uint8 .npy."""
import sys
import numpy as np
rng = np.random.default_rng(7)
H = W = 256
yy, xx = np.meshgrid(np.linspace(0, 1, H), np.linspace(0, 1, W), indexing="ij")
img = 0.35 + 0.3 * xx - 0.15 * yy
for _ in range(12):
    cy, cx, s, a = rng.uniform(0, 1), rng.uniform(0, 1), rng.uniform(0.03, 0.2), rng.uniform(-0.35, 0.35)
    img += a * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * s * s))
img += 0.2 * (xx + 0.5 * yy > 0.9) - 0.15 * ((xx - 0.3) ** 2 + (yy - 0.6) ** 2 < 0.02)
img += rng.normal(scale=1.5 / 255, size=img.shape)
np.save(sys.argv[1], np.uint8(np.round(np.clip(img, 0, 1) * 255))[..., None])
