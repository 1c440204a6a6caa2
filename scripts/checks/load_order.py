import sys
sys.path.insert(0, "/root/repo")
from pylbl_amd.engine import Engine
e = Engine(0)
import torch
x = torch.zeros(4, device="cuda")
print("engine first, torch second: ok", torch.cuda.device_count(), float(x.sum()))
