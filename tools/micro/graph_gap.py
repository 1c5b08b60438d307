import torch, time
x = torch.zeros(256, device="cuda")
s = torch.cuda.Stream()
def chain(n):
    for _ in range(n): x.add_(1.0)
for n in (1, 60, 120, 240):
    with torch.cuda.stream(s):
        chain(n); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            chain(n)
        for _ in range(5): g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50): g.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 50
    print(f"graph of {n} dependent tiny kernels: {dt*1e6:.1f} us per replay = {dt*1e6/n:.2f} us per node")
