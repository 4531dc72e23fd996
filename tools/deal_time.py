import numpy as np, sys, os
sys.path.insert(0, '.')
from tests.test_deal_gpu import deal
rng = np.random.default_rng(1)
G = 7808
keys = np.where(rng.random(G) < 0.7, rng.integers(0, 255, G), 255).astype(np.uint32)
deal(keys)
