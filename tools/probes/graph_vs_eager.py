"""debug probe: run-to-run spread of losses / parameters / BN running statistics, eager vs hipGraph replay"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import torch
from test_graph_gpu import _run
dev = torch.device("cuda", 0)
torch.set_printoptions(precision=3, linewidth=200)
def d(a, b, floor=None):
    return float(((a - b).abs() / a.abs().clamp_min(floor)).max()) if floor else float((a - b).abs().max())
for dtype, tie in ((torch.float32, False), (torch.float32, True)):
    E = [_run(dev, False, 6, dtype, tie) for _ in range(3)]
    G = [_run(dev, True, 6, dtype, tie) for _ in range(3)]
    print(dtype, "tie", tie)
    for name, A, B in (("e-e", E, E), ("g-g", G, G), ("e-g", E, G)):
        pairs = [(i, j) for i in range(3) for j in range(3) if (A is not B or i < j)]
        print("  %s loss %s  params %s  running %s" % (
            name, ["%.1e" % d(A[i][0], B[j][0], 1e-9) for i, j in pairs],
            ["%.1e" % d(A[i][1], B[j][1]) for i, j in pairs], ["%.1e" % d(A[i][2], B[j][2], 1.0) for i, j in pairs]))
