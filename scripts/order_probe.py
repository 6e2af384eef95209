import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pecos_amd import clib
print("devices seen by the library:", clib.device_count(), "torch imported:", "torch" in sys.modules)
import torch
print("torch after the library: cuda available =", torch.cuda.is_available(), torch.zeros(4, device="cuda").sum().item())
