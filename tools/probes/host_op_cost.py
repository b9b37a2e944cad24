"""host-side issue cost (us per call) of the main op wrappers, GPU work tiny so the queue never fills"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from fsnet_amd.hip import ops
from fsnet_amd.hip.conv import ConvOp
dev = torch.device("cuda", 0)
dt = torch.bfloat16
op = ConvOp(64, 64, 3, 3, 1, 1, dt, dev)
op.pack(torch.randn(64, 64, 3, 3, device=dev))
x = torch.randn(1, 8, 16, 64, device=dev).to(dt)
y = torch.empty(1, 8, 16, 64, device=dev, dtype=dt)
dw = torch.zeros(64, 64, 3, 3, device=dev)
stats = torch.zeros(8, 2, 64, dtype=torch.float64, device=dev)


def t(name, fn, n=3000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    el = time.perf_counter() - t0
    torch.cuda.synchronize()
    print("%-28s %6.2f us/call" % (name, el / n * 1e6))


t("conv forward (out given)", lambda: op.forward(x, out=y, stats=stats))
t("conv forward (alloc out)", lambda: op.forward(x, stats=stats))
t("conv dgrad", lambda: op.dgrad(y, 8, 16, out=x))
t("conv wgrad", lambda: op.wgrad(y, x, dw))
t("torch.empty", lambda: torch.empty(1, 8, 16, 64, device=dev, dtype=dt))
t("x.data_ptr()", lambda: x.data_ptr())
from fsnet_amd.hip.binding import stream_ptr, lib
t("stream_ptr()", lambda: stream_ptr())
t("raw lib call (counter_incr)", lambda: lib.fs_counter_incr(stats.data_ptr(), stream_ptr()))
