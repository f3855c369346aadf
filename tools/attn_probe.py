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

# training pair (round 3): fused forward (+ log-sum-exp, dropout) and the recomputing backward vs the unfused Function
print("training (dropout 0.2): forward + backward")
for (B, T, H, dk) in [(32, 800, 2, 128), (32, 128, 2, 128)]:
    C = H * dk
    base = [torch.randn(B, T, C, device=dev) for _ in range(3)]
    dout = torch.randn(B, T, C, device=dev)
    klen = torch.full((B,), T, device=dev, dtype=torch.int64)
    res = {}
    for fused in (True, False):
        ops._FUSED_ATTN_TRAIN = fused
        q, k, v = (t.clone().requires_grad_(True) for t in base)
        def step():
            o = ops.AttentionFn.apply(q, k, v, klen, H, 0.2, 99, 3)
            o.backward(dout)
            q.grad = k.grad = v.grad = None
        res[fused] = timeit(step, reps=10)
    ops._FUSED_ATTN_TRAIN = True
    o, lse = K.attn_train_fwd(base[0], base[1], base[2], klen, H, 0.2, 99, 3)
    tfw = timeit(lambda: K.attn_train_fwd(base[0], base[1], base[2], klen, H, 0.2, 99, 3))
    tbw = timeit(lambda: K.attn_train_bwd(base[0], base[1], base[2], o, lse, dout, klen, H, 0.2, 99, 3))
    fl = 4.0 * B * H * T * T * dk
    print(f"B={B} T={T} H={H} dk={dk}: fused fwd {tfw:7.1f} us ({fl/tfw/1e6:4.0f} TF), fused bwd (dQ + dK/dV kernels) {tbw:7.1f} us "
          f"({3.5*fl/tbw/1e6:4.0f} TF incl. recompute); Function fwd+bwd fused {res[True]:7.1f} us vs unfused {res[False]:7.1f} us; "
          f"(T x T) tensors not written: 3 x {B*H*T*T*4/1e6:.0f} MB")
