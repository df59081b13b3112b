"""Load libedison_hip BEFORE torch and use both: guards the HIP-runtime sharing in edison_amd/_lib.py."""
import sys
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from edison_amd.context import Context
assert "torch" not in sys.modules
c = Context(0)
m = c.mfcc_q15(np.zeros(2048, np.int16))
import torch
t = torch.zeros(4, device="cuda") + 1
assert float(t.sum()) == 4.0 and m.shape == (2, 32)
print("ok: library first, torch second, both see the GPU")
