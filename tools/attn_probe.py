"""Fused (flash-style) attention forward vs the unfused path (two batched GEMMs + softmax kernel) at the decoder size."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optispeech_amd import ops, precision, kernels as K
precision.set_precision("bf16")
dev = "cuda"
def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for (B, T, H, dk) in [(32, 800, 2, 128), (32, 128, 2, 128), (64, 250, 2, 128)]:
    C = H * dk
    q, k, v = (torch.randn(B, T, C, device=dev) for _ in range(3))
    klen = torch.full((B,), T, device=dev, dtype=torch.int64)
    with torch.no_grad():
        tf = timeit(lambda: K.attn_fused_fwd(q, k, v, klen, H))
        os.environ["X"] = "1"
        ops._FUSED_ATTN = False
        tu = timeit(lambda: ops.AttentionFn.apply(q, k, v, klen, H, 0.0, 0, 0))
        ops._FUSED_ATTN = True
    fl = 4.0 * B * H * T * T * dk
    print(f"B={B} T={T} H={H} dk={dk}: fused {tf:7.1f} us ({fl/tf/1e6:5.0f} TFLOP/s), unfused {tu:7.1f} us; scores not written: {B*H*T*T*4/1e6:.0f} MB")
