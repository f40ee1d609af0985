"""Round 5: kernel-level view of torch.linalg.eigh (rocSOLVER syevd) on a 206 x 206 graded matrix -- run under rocprofv3 --kernel-trace --stats."""
import torch
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
r = 206
A = torch.randn(r, r, device=dev, dtype=torch.float64, generator=g)
w = torch.logspace(0, -7, r, device=dev, dtype=torch.float64)
Q, _ = torch.linalg.qr(A)
T = (Q * w) @ Q.T
for _ in range(10):
    torch.linalg.eigh(T)
torch.cuda.synchronize()
