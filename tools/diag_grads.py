"""GPU diagnostic: per-parameter gradient agreement with the oracle (not a test)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_step_gpu import _setup, _cos

def run(kind, dims, B, T, seed, dtype=torch.bfloat16):
    O, E, ocfg, ecfg, params, batch, eps = _setup(kind, dims, B, T, seed)
    gpu = torch.device("cuda", 0)
    ot = O.OracleTrainer(ocfg, params, lr=1e-3)
    ref = ot.step(batch, torch.from_numpy(eps))
    store = E.ParamStore(ecfg, gpu, dtype, params_np=params)
    plan = E.StepPlan(store, B, T, lr=1e-3, want_probs=True)
    plan.load_batch(batch["x"], batch["seq_lens"], batch["classes"], batch["labels"], eps)
    plan._tick_adam = True  # forward() then clears the gradient bucket (and advances the Adam step counter)
    plan.forward(); plan.losses(True); plan.backward(); torch.cuda.synchronize()
    g = store.to_numpy("g")
    print(f"--- {kind} dims={dims} B={B} T={T} gscale={plan.gscale}")
    for name, rg in ref["grads"].items():
        rg = rg.numpy() * plan.gscale
        c = _cos(g[name], rg)
        flag = "" if c > 0.98 else "   <<<<<<"
        print(f" {name:42s} cos {c:8.5f} |ref|max {np.abs(rg).max():9.3g} |got|max {np.abs(g[name]).max():9.3g}{flag}")
run("pianoroll", (40, 40, 2, 16, 64, 2, 2, 32, 1, 2), 5, 19, 12)
run("token", (10, 10, 3, 16, 32, 1, 2, 32, 1, 2), 3, 5, 7)
