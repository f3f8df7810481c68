"""Scratch: TNMLRegressor.fit at the headline configuration (60000 x 16, degree 5, bond 32), wall time per swipe."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tensornetworksfork_amd.models.tnml import TNMLRegressor
rng = np.random.RandomState(0)
X = rng.uniform(-1, 1, (60000, 16))
y = np.sin(X).sum(1, keepdims=True) + 0.01 * rng.randn(60000, 1)
for swipes in (1, 2):
    m = TNMLRegressor(r=32, seed=42, basis='polynomial', degree=5, num_swipes=swipes, early_stopping=10000, eps_start=1.0,
                      eps_decay=0.5, method='ridge_cholesky', batch_size=512)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.fit(X, y, validation_split=0.1) if 'validation_split' in TNMLRegressor.fit.__code__.co_varnames else m.fit(X, y)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"num_swipes={swipes}: fit {1e3 * (t1 - t0):.1f} ms")
