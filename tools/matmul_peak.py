import torch, time
dev = torch.device("cuda:0")
for n in (4096, 8192):
    for fill in ("rand", "zero"):
        a = (torch.randn(n, n, device=dev) if fill == "rand" else torch.zeros(n, n, device=dev)).bfloat16()
        b = (torch.randn(n, n, device=dev) if fill == "rand" else torch.zeros(n, n, device=dev)).bfloat16()
        for _ in range(3): c = a @ b
        torch.cuda.synchronize(); t = time.time()
        for _ in range(20): c = a @ b
        torch.cuda.synchronize(); dt = (time.time() - t) / 20
        print(f"hipBLASLt bf16 {n}^3 {fill}: {2*n**3/dt/1e12:.0f} TFLOP/s")
