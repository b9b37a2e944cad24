"""does hipGraph capture survive per-layer fork to a companion stream? (variants run in subprocesses)"""
import subprocess, sys
import torch

def run(variant, layers):
    dev = torch.device("cuda", 0)
    main = torch.cuda.Stream()
    ws = torch.cuda.Stream()
    side = torch.cuda.Stream()
    ws2 = torch.cuda.Stream()
    x = torch.randn(256, 256, device=dev)
    outs = [torch.zeros(256, 256, device=dev) for _ in range(layers)]
    outs2 = [torch.zeros(256, 256, device=dev) for _ in range(layers)]
    def chain(cur, comp, x, outs):
        y = x
        for i in range(layers):
            y = torch.tanh(y @ x)
            if variant in ("A", "B", "C") or (variant == "D" and cur is side) or (variant == "F" and cur is not side) \
                    or (variant == "G"):
                comp.wait_stream(cur)
                with torch.cuda.stream(comp):
                    outs[i].add_(y.t() @ y)
                if variant == "B":
                    cur.wait_stream(comp)
        if variant != "G":
            cur.wait_stream(comp)
        return y
    def step():
        if variant in ("C", "D", "F", "G"):
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                chain(side, ws2, x, outs2)
        y = chain(torch.cuda.current_stream(), ws, x, outs)
        if variant in ("C", "D", "F", "G"):
            torch.cuda.current_stream().wait_stream(side)
        if variant == "G":       # all companions joined directly into the capture stream, at the very end
            torch.cuda.current_stream().wait_stream(ws)
            torch.cuda.current_stream().wait_stream(ws2)
        return y
    with torch.cuda.stream(main):
        step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=main):
        y = step()
    g.replay(); torch.cuda.synchronize()
    print("variant", variant, "layers", layers, "ok", float(y.sum()))

if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1], int(sys.argv[2]))
    else:
        for v in ("C", "D", "F", "G"):
            for L in (1, 8):
                r = subprocess.run([sys.executable, __file__, v, str(L)], capture_output=True, text=True)
                print(v, L, "rc", r.returncode, r.stdout.strip()[-60:], r.stderr.strip()[-200:].replace("\n", " | ") if r.returncode else "")
