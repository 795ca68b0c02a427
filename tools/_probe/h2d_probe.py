import torch, time
dev = torch.device("cuda:0")
n = 256 * 71680
for nstreams in (1, 2, 4):
    hs = [torch.empty(n // nstreams, dtype=torch.uint8).pin_memory() for _ in range(2 * nstreams)]
    ds = [torch.empty(n // nstreams, dtype=torch.uint8, device=dev) for _ in range(2 * nstreams)]
    streams = [torch.cuda.Stream(dev) for _ in range(nstreams)]
    def run(iters):
        for _ in range(iters):
            for i, s in enumerate(streams):
                with torch.cuda.stream(s):
                    ds[2 * i].copy_(hs[2 * i], non_blocking=True)
                    ds[2 * i + 1].copy_(hs[2 * i + 1], non_blocking=True)
        torch.cuda.synchronize()
    run(3)
    t = time.perf_counter(); run(20); dt = (time.perf_counter() - t) / 20
    print(f"{nstreams} stream(s): {2 * n / 1e6:.1f} MB per step in {dt * 1e3:.3f} ms = {2 * n / dt / 1e9:.1f} GB/s")
