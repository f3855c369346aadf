import os, sys, torch
sys.path.insert(0, "/root/repo")
from optispeech_amd import kernels as K
dev="cuda"
M,N,Kd=13056,1024,5120
a=torch.randn(M,Kd,device=dev).bfloat16(); w=torch.randn(N,Kd,device=dev).bfloat16()
out=torch.empty(M,N,device=dev,dtype=torch.bfloat16)
for _ in range(3):
    K.conv_gemm_bf16(a,w,N,M=M,Trows=M,Tin=M,cin=Kd,out=out,out_bf16=True)
torch.cuda.synchronize()
