cd /root/repo
python -m pytest tests/test_conv_fold_gpu.py -q -m gpu --no-header -p no:cacheprovider -k "persistent" 2>&1 | grep -E "^E|passed|failed" | head -20
python - <<'PY'
import torch
from fsnet_amd.hip.conv import ConvOp
dev=torch.device("cuda:0")
for Ci,Co,N,H,W in [(16,16,12,192,640),(32,16,12,96,320),(16,16,12,194,642)]:
    op=ConvOp(Ci,Co,3,3,1,1,torch.bfloat16,dev)
    print(Ci,Co,H,W, op.plan_3x3(N,H,W,forward=True), op.plan_3x3(N,H,W,forward=False))
PY
