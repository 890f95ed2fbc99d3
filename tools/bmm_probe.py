import torch, time
dev = "cuda:0"
for (b, m, k, n) in [(1024, 256, 200, 200), (1024, 200, 256, 200), (4, 256, 51200, 200)]:
    A = torch.randn(b, m, k, device=dev); B = torch.randn(b, k, n, device=dev)
    for _ in range(3): C = torch.bmm(A, B)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): C = torch.bmm(A, B)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    print("bmm fp32 [%d x %d x %d x %d]: %.1f us  %.1f TFLOP/s" % (b, m, k, n, dt * 1e6, 2.0 * b * m * k * n / dt / 1e12))
