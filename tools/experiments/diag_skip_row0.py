"""which probabilities differ between MST_SKIP_ROW0=0 and 1 (engine only, configs[2] shape)?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_step_gpu import _setup
dims = tuple(int(a) for a in sys.argv[1].split(",")) if len(sys.argv) > 1 else (2048, 2048, 2, 256, 256, 2, 8, 128, 1, 8)
B, T = 64, 256
O, E, ocfg, ecfg, params, batch, eps = _setup("pianoroll", dims, B, T, 2048)
gpu = torch.device("cuda", 0)
res = {}
for skip in ("0", "1"):
    os.environ["MST_SKIP_ROW0"] = skip
    store = E.ParamStore(ecfg, gpu, torch.bfloat16, params_np=params)
    plan = E.StepPlan(store, B, T, lr=3e-4, clip_gradient=1.0, want_probs=True)
    plan.load_batch(batch["x"], batch["seq_lens"], batch["classes"], batch["labels"], eps)
    plan.step_kernels(True)
    torch.cuda.synchronize()
    L = plan.dec[0]
    res[skip] = {k: getattr(L, k).float().cpu().view(B, T + 1, -1) for k in ("att", "h1", "x1", "h2", "x2")}
    res[skip]["probs"] = plan.probs.float().cpu()[:, :dims[1]].reshape(B, T, -1)
    print("skip", skip, "skip_row0 =", plan.skip_row0, "fuse_bce", plan.fuse_bce)
for k in res["0"]:
    a, b = res["0"][k], res["1"][k]
    if k != "probs":
        a, b = a[:, 1:], b[:, 1:]
    d = (a - b).abs()
    bad = (d > 0.05).any(-1)
    print(k, "max diff", d.max().item(), "rows off", int(bad.sum()), "first", torch.nonzero(bad)[:6].tolist())
