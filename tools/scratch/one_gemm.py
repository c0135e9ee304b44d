import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from musicstyletransfer_amd import ops as o
BF=torch.bfloat16; dev=torch.device("cuda",0)
M,N,K=16384,1024,256
A=torch.randn(M,K,device=dev).to(BF); W=(torch.randn(N,K,device=dev)*0.05).to(BF); C=torch.zeros(M,N,dtype=BF,device=dev)
bias=torch.randn(N,device=dev)
for _ in range(10): o.gemm_nt(A,W,C,bias=bias,act=o.ACT_RELU)
torch.cuda.synchronize()
