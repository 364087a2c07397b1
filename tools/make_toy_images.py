import numpy as np, os, sys
from PIL import Image
d=sys.argv[1]; os.makedirs(d, exist_ok=True)
rng=np.random.RandomState(1)
for i in range(16): Image.fromarray(rng.randint(0,256,size=(80,72,3),dtype=np.uint8)).save(f"{d}/{i:03d}.png")
