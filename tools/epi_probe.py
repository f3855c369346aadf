import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from optispeech_amd import kernels as K
dev="cuda"
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
M,N=13056,1024
out=torch.empty(M,N,device=dev,dtype=torch.bfloat16)
print("fill bf16 %.1f us"%t(lambda: out.fill_(1.0)))
for Kd in (64,128,256,1024,5120):
    a=torch.randn(M,Kd,device=dev).bfloat16(); w=torch.randn(N,Kd,device=dev).bfloat16()
    print("K=%d: %.1f us (out preallocated)"%(Kd,t(lambda: K.conv_gemm_bf16(a,w,N,M=M,Trows=M,Tin=M,cin=Kd,out=out,out_bf16=True))))
    outf=torch.empty(M,N,device=dev)
    print("K=%d: %.1f us f32 out"%(Kd,t(lambda: K.conv_gemm_bf16(a,w,N,M=M,Trows=M,Tin=M,cin=Kd,out=outf))))
