import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from musicstyletransfer_amd import ops as o
BF=torch.bfloat16; dev=torch.device("cuda",0)
M=16384; D=256
dh, a = torch.randn(M, D, device=dev).to(BF), torch.randn(M, 4 * D, device=dev).to(BF)
dpre, x1 = torch.randn(M, 4 * D, device=dev).to(BF), torch.randn(M, D, device=dev).to(BF)
dqkv = torch.randn(M, 3 * D, device=dev).to(BF)
g = [torch.zeros(D, 4 * D, device=dev), torch.zeros(4 * D, D, device=dev), torch.zeros(D, D, device=dev), torch.zeros(3 * D, D, device=dev)]
b = [torch.zeros(D, device=dev), torch.zeros(4 * D, device=dev), torch.zeros(D, device=dev), torch.zeros(3 * D, device=dev)]
for _ in range(6):
    o.gemm_wgrad_batch([o.wgrad_problem(dh, a, g[0], b[0]), o.wgrad_problem(dpre, x1, g[1], b[1]),
                        o.wgrad_problem(dh, x1, g[2], b[2]), o.wgrad_problem(dqkv, x1, g[3], b[3])])
torch.cuda.synchronize()
