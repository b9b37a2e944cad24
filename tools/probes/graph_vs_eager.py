"""debug probe: per-step loss differences eager vs hipGraph replay (bf16, tie noise on)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import torch
from test_graph_gpu import _run
from fsnet_amd.engine.runtime import RT
dev = torch.device("cuda", 0)
torch.set_printoptions(precision=3, linewidth=200)
for overlap in (True, False):
    RT.overlap = overlap
    for dtype in (torch.bfloat16,):
        e1 = _run(dev, False, 8, dtype, True)
        e2 = _run(dev, False, 8, dtype, True)
        g1 = _run(dev, True, 8, dtype, True)
        g2 = _run(dev, True, 8, dtype, True)
        rel = lambda a, b: ((a[0] - b[0]).abs() / a[0].abs())
        print("overlap", overlap, dtype)
        print(" e1-e2", rel(e1, e2))
        print(" e1-g1", rel(e1, g1))
        print(" g1-g2", rel(g1, g2))
        print(" params e1-e2 %.3e e1-g1 %.3e g1-g2 %.3e" % ((e1[1]-e2[1]).abs().max(), (e1[1]-g1[1]).abs().max(), (g1[1]-g2[1]).abs().max()))
