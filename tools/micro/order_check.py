import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lasercalib_amd import _native
print("native devices:", _native.device_count())
import torch
print("torch cuda:", torch.cuda.is_available(), torch.cuda.device_count())
x = torch.zeros(4, device="cuda"); print(x.sum().item())
