echo "--- mine first"; python - <<'PY' 2>&1 | tail -3
import sys; sys.path.insert(0, '.')
from fluctus_amd import device
device.lib()
import torch
print("torch sees", torch.cuda.is_available())
try:
    device.HipContext(1024); print("ctx ok")
except Exception as e: print("ctx FAIL", e)
PY
echo "--- torch first"; python - <<'PY' 2>&1 | tail -3
import sys; sys.path.insert(0, '.')
import torch
print("torch sees", torch.cuda.is_available())
from fluctus_amd import device
try:
    device.HipContext(1024); print("ctx ok")
except Exception as e: print("ctx FAIL", e)
PY
echo "--- no torch"; python - <<'PY' 2>&1 | tail -3
import sys; sys.path.insert(0, '.')
from fluctus_amd import device
try:
    device.HipContext(1024); print("ctx ok")
except Exception as e: print("ctx FAIL", e)
PY
