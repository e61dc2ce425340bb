import time, numpy as np, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import overiva_amd as oa
from oracle.overiva_oracle import synth_mixture
X = synth_mixture(160, 2049, 8, 2, seed=9).astype(np.complex128)
oa.set_precision(sys.argv[1] if len(sys.argv) > 1 else "precise")
oa.ogive(X, n_iter=400, tol=0.0)
