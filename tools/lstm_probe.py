"""LSTM recurrence timing at the BASELINE shapes (32 utterances, T = 800 mel frames / 128 text tokens, H = 256)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optispeech_amd import kernels as K
dev = "cuda"
for (B, T, H) in [(32, 800, 256), (32, 128, 256)]:
    gx = torch.randn(B, T, 4 * H, device=dev) * 0.5
    whh = torch.randn(4 * H, H, device=dev) * 0.05
    for _ in range(2):
        hs, gates, cs = K.lstm_fwd(gx, whh)
    torch.cuda.synchronize()
    a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    a.record(); hs, gates, cs = K.lstm_fwd(gx, whh); b.record()
    dg = K.lstm_bwd(torch.randn_like(hs), gates, cs, whh); c.record(); torch.cuda.synchronize()
    print(f"B={B} T={T} H={H}: forward {a.elapsed_time(b):.2f} ms ({a.elapsed_time(b)/T*1e3:.2f} us/step), backward {b.elapsed_time(c):.2f} ms ({b.elapsed_time(c)/T*1e3:.2f} us/step)")
