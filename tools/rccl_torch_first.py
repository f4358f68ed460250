import torch, sys
sys.path.insert(0, '/root/repo')
torch.cuda.set_device(0)
from fluid_simulation_amd import _lib
_lib.check(_lib.lib().fs_comm_selftest())
print("selftest ok with torch imported first")
import subprocess, os
maps = open('/proc/self/maps').read()
print(sorted(set(l.split()[-1] for l in maps.splitlines() if 'librccl' in l or 'libamdhip64' in l)))
