"""Counts the runs (three eager Adam steps of a small piano-roll model whose tails have riders) that end away from the majority.
usage: [ENV=...] diag_sporadic.py [reps] [steps]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
from test_step_gpu import _setup
gpu = torch.device("cuda", 0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
O, E, ocfg, ecfg, params, batch, eps = _setup("pianoroll", (64, 64, 2, 16, 128, 2, 4, 128, 1, 4), 4, 128, 67, ragged=(os.environ.get("DIAG_RAGGED", "1") != "0"))
ws = []
for rep in range(reps):
    store = E.ParamStore(ecfg, gpu, torch.bfloat16, params_np=params)
    plan = E.StepPlan(store, 4, 128, lr=1e-2)
    plan.load_batch(batch["x"], batch["seq_lens"], batch["classes"], batch["labels"], eps)
    for _ in range(steps):
        plan.step_kernels(True)
    torch.cuda.synchronize()
    ws.append(store.w.cpu().numpy().copy())
    ride = (plan.ride, plan._ride_fwd, plan._ride_bwd, plan._tail_shadows is not None)
    del plan, store
d = np.array([[(np.abs(a - b) > 2e-5).mean() for b in ws] for a in ws])
med = np.median(d, axis=1)
print("env", {k: v for k, v in os.environ.items() if k.startswith("MST_")}, "ride", ride)
print("fraction of weights away from the other runs (median per run):", " ".join(f"{m:.3f}" for m in med), " -> bad runs:", int((med > 0.1).sum()), "of", reps, flush=True)
