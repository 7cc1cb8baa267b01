echo "== env"; env | grep -i -E "hip|rocr|cuda|hsa|gpu" 
echo "== torch alone"; python -c "
import torch; print(torch.__version__, torch.version.hip, torch.cuda.is_available(), torch.cuda.device_count())
x=torch.ones(4,device='cuda'); print(x.sum().item())
import os
print([l.strip().split()[-1] for l in open('/proc/self/maps') if 'amdhip' in l or 'hsa-runtime' in l][:6])
" 2>&1 | tail -5
echo "== lib first then torch"; python -c "
import sys; sys.path.insert(0,'.')
from fastga_amd import device as D
d=D.Device(0)
import torch
print('avail',torch.cuda.is_available(), torch.cuda.device_count())
try:
    x=torch.ones(4,device='cuda'); print(x.sum().item())
except Exception as e: print('ERR',e)
print(sorted(set(l.strip().split()[-1] for l in open('/proc/self/maps') if 'amdhip' in l or 'hsa-runtime' in l)))
" 2>&1 | tail -6
echo "== torch first then lib"; python -c "
import sys; sys.path.insert(0,'.')
import torch
x=torch.ones(4,device='cuda'); print(x.sum().item())
from fastga_amd import device as D
try:
    d=D.Device(0); print('lib ok')
except Exception as e: print('ERR',e)
print(sorted(set(l.strip().split()[-1] for l in open('/proc/self/maps') if 'amdhip' in l or 'hsa-runtime' in l)))
" 2>&1 | tail -6
ls /usr/local/lib/python3.10/dist-packages/torch/lib | grep -E "amdhip|hsa|rccl" 
