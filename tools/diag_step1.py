"""GPU diagnostic: why does step 1 differ? (not a test)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from collections import OrderedDict
from tests.test_step_gpu import _setup

def run(kind, dims, B, T, seed, dtype):
    O, E, ocfg, ecfg, params, batch, eps = _setup(kind, dims, B, T, seed)
    gpu = torch.device("cuda", 0)
    lr = 1e-3
    ot = O.OracleTrainer(ocfg, params, lr=lr)
    store = E.ParamStore(ecfg, gpu, dtype, params_np=params)
    plan = E.StepPlan(store, B, T, lr=lr, want_probs=True)
    plan.load_batch(batch["x"], batch["seq_lens"], batch["classes"], batch["labels"], eps)
    ref0 = ot.step(batch, torch.from_numpy(eps)); plan.step_kernels(True); torch.cuda.synchronize()
    print(f"--- {kind} {dtype} step0 ELBO gpu {plan.total.mean().item():.5f} ref {ref0['loss'].mean().item():.5f}")
    w = store.to_numpy("w")
    tot_flip = tot_n = 0
    for name, p in ot.P.items():
        dw_ref = p.detach().numpy() - params[name]
        dw_gpu = w[name] - params[name]
        flips = (np.abs(dw_gpu - dw_ref) > 1.0 * lr).sum()
        zero_ref = (dw_ref == 0).sum()
        moved_gpu_but_not_ref = ((dw_ref == 0) & (np.abs(dw_gpu) > 0.5 * lr)).sum()
        tot_flip += flips; tot_n += dw_ref.size
        if flips > 0.02 * dw_ref.size or moved_gpu_but_not_ref:
            print(f"  {name:40s} n {dw_ref.size:7d} flips {flips:6d} ref-unmoved {zero_ref:6d} gpu-moved-where-ref-didnt {moved_gpu_but_not_ref}")
    print(f"  total flips {tot_flip}/{tot_n}")
    # oracle forward with the GPU's weights
    P_gpu = O.to_torch_params(w, requires_grad=False)
    loss_g, recon_g, kl_g, *_ = O.step_losses(P_gpu, ocfg, batch, torch.from_numpy(eps))
    ref1 = ot.step(batch, torch.from_numpy(eps)); plan.step_kernels(True); torch.cuda.synchronize()
    print(f"  step1 ELBO: gpu {plan.total.mean().item():.5f} | oracle fwd on GPU weights {loss_g.mean().item():.5f} | oracle {ref1['loss'].mean().item():.5f}")
run("pianoroll", (40, 40, 2, 16, 64, 2, 2, 32, 1, 2), 5, 19, 12, torch.bfloat16)
run("pianoroll", (40, 40, 2, 16, 64, 2, 2, 32, 1, 2), 5, 19, 12, torch.float16)
