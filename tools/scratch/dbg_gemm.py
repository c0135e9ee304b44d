import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from musicstyletransfer_amd import ops as o
from tools.bench_gemm import timeit
BF=torch.bfloat16; dev=torch.device("cuda",0)
for (M,N,K) in [(16384,1024,256),(32768,1024,256),(16384,256,1024),(16384,256,256)]:
    A=torch.randn(M,K,device=dev).to(BF); W=(torch.randn(N,K,device=dev)*0.05).to(BF); C=torch.zeros(M,N,dtype=BF,device=dev)
    bias=torch.randn(N,device=dev)
    res=[]
    for dbg in (0,1,2,4,3,7):
        os.environ["MST_GEMM_DBG"]=str(dbg)
        res.append((dbg, timeit(lambda: o.gemm_nt(A,W,C,bias=bias,act=o.ACT_RELU))))
    print(M,N,K, " ".join(f"dbg{d}={t:.1f}us" for d,t in res))
