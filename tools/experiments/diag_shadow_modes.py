"""Which shadow-refresh placement moves the weights? Three Adam steps of a small piano-roll model whose tails have riders, under
MST_SHADOW_RIDE=0 (launch of its own), MST_SHADOW_TAIL=0 (the step's first launch), default (forward tail's riders); repeated."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from test_step_gpu import _setup
gpu = torch.device("cuda", 0)
O, E, ocfg, ecfg, params, batch, eps = _setup("pianoroll", (64, 64, 2, 16, 128, 2, 4, 128, 1, 4), 4, 128, 67)
allm = {"own": {"MST_SHADOW_RIDE": "0"}, "begin": {"MST_SHADOW_TAIL": "0"}, "tail": {}}
order = sys.argv[1].split(",") if len(sys.argv) > 1 else ["own", "begin", "tail"]
modes = {k: allm[k] for k in order}
res = {}
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    for name, env in modes.items():
        for k, v in env.items():
            os.environ[k] = v
        try:
            store = E.ParamStore(ecfg, gpu, torch.bfloat16, params_np=params)
            plan = E.StepPlan(store, 4, 128, lr=1e-2)
            plan.load_batch(batch["x"], batch["seq_lens"], batch["classes"], batch["labels"], eps)
            ws = []
            for _ in range(3):
                plan.step_kernels(True)
                torch.cuda.synchronize()
                ws.append(store.w.cpu().numpy().copy())
            res[(name, rep)] = ws
            print(name, rep, "status", store.step_status.cpu().tolist()[:4], "tail shadows", plan._tail_shadows is not None, flush=True)
        finally:
            for k in env:
                os.environ.pop(k, None)
ref = res[(order[0], 0)]
for key, ws in res.items():
    print(key, " ".join(f"step{i + 1}: {(np.abs(w - r) > 2e-5).mean():.4f}" for i, (w, r) in enumerate(zip(ws, ref))))
