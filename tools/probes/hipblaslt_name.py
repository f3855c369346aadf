import torch
a = torch.randn(16320, 5120, device="cuda").bfloat16(); w = torch.randn(1024, 5120, device="cuda").bfloat16()
for _ in range(5):
    torch.matmul(a, w.t())
a = torch.randn(13056, 5120, device="cuda").bfloat16()
for _ in range(5):
    torch.matmul(a, w.t())
torch.cuda.synchronize()
