"""dependency latency of a captured chain: N one-thread kernels (fs_debug_timestamp) in a row on one stream, replayed;
and the same with every second kernel on a second stream (fork/join per node)"""
import sys, time
sys.path.insert(0, '.')
import torch
from fsnet_amd.hip.binding import lib, check
dev = torch.device('cuda:0')
buf = torch.zeros(4096, dtype=torch.int64, device=dev)
N = 400
def chain(two_streams):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(); s2 = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for i in range(N):
                if two_streams and i % 2:
                    s2.wait_stream(s)
                    check(lib.fs_debug_timestamp(buf.data_ptr() + 8 * i, s2.cuda_stream))
                    s.wait_stream(s2)
                else:
                    check(lib.fs_debug_timestamp(buf.data_ptr() + 8 * i, s.cuda_stream))
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    t = buf.cpu().tolist()
    d = sorted(t[i + 1] - t[i] for i in range(N - 1))
    print("two_streams=%d: %.2f us per node wall (host %.2f us per node); device-clock gap median %.2f us  p90 %.2f" % (
        two_streams, (t2 - t0) / 10 / N * 1e6, (t1 - t0) / 10 / N * 1e6, d[len(d) // 2] / 100.0, d[int(len(d) * .9)] / 100.0))
chain(False); chain(True)
