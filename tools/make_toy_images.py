"""Toy folder dataset for smoke runs of train.py: smooth random colour blobs (n images of size x size PNGs)."""
import os
import sys

import numpy as np
from PIL import Image

d, n, size = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 16, int(sys.argv[3]) if len(sys.argv) > 3 else 80
os.makedirs(d, exist_ok=True)
rng = np.random.RandomState(1)
yy, xx = np.mgrid[0:size, 0:size].astype(np.float32) / size
for i in range(n):
    img = np.zeros((size, size, 3), np.float32)
    for _ in range(6):
        cx, cy, s = rng.rand(), rng.rand(), 0.05 + 0.25 * rng.rand()
        img += np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))[..., None] * rng.rand(3)
    img = img / img.max() * 255
    Image.fromarray(img.astype(np.uint8)).save(f"{d}/{i:04d}.png")
