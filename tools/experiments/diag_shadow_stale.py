"""After two eager steps the backward-only transposed shadows must equal the weights the first step's optimizer left (w1): which
copy do they hold when the run goes wrong? usage: diag_shadow_stale.py [begin|tail] [reps]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
from test_step_gpu import _setup
gpu = torch.device("cuda", 0)
mode = sys.argv[1] if len(sys.argv) > 1 else "begin"
if mode == "begin":
    os.environ["MST_SHADOW_TAIL"] = "0"
O, E, ocfg, ecfg, params, batch, eps = _setup("pianoroll", (64, 64, 2, 16, 128, 2, 4, 128, 1, 4), 4, 128, 67)
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 10):
    store = E.ParamStore(ecfg, gpu, torch.bfloat16, params_np=params)
    plan = E.StepPlan(store, 4, 128, lr=1e-2)
    plan.load_batch(batch["x"], batch["seq_lens"], batch["classes"], batch["labels"], eps)
    w0 = store.w.clone()
    plan.step_kernels(True)
    w1 = store.w.clone()
    plan.step_kernels(True)
    torch.cuda.synchronize()
    emb = ["encoder.embedding.weight", "decoder.embedding.weight"]
    out = []
    for n in store.t_specs:
        if n in emb:
            continue
        so, r, c = store.t_specs[n]
        got = store.t(n)[:, :r]
        e1 = bool(torch.equal(got, w1[so: so + r * c].view(r, c).t().to(store.act_dtype)))
        e0 = bool(torch.equal(got, w0[so: so + r * c].view(r, c).t().to(store.act_dtype)))
        if not e1:
            frac1 = (got != w1[so: so + r * c].view(r, c).t().to(store.act_dtype)).float().mean().item()
            out.append(f"{n}: w1 {e1} w0 {e0} mismatch {frac1:.3f}")
    print(rep, "status", store.step_status.cpu().tolist()[:3], "OK" if not out else f"{len(out)} stale: " + "; ".join(out[:4]), flush=True)
