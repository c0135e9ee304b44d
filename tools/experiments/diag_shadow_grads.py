"""Repeated two-step runs of one mode: which parameters' step-2 gradients differ between a good and a bad run? usage: diag_shadow_grads.py [begin|tail|own] [reps]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
from test_step_gpu import _setup
gpu = torch.device("cuda", 0)
mode = sys.argv[1] if len(sys.argv) > 1 else "begin"
if mode == "begin":
    os.environ["MST_SHADOW_TAIL"] = "0"
if mode == "own":
    os.environ["MST_SHADOW_RIDE"] = "0"
O, E, ocfg, ecfg, params, batch, eps = _setup("pianoroll", (64, 64, 2, 16, 128, 2, 4, 128, 1, 4), 4, 128, 67)
runs = []
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 10):
    store = E.ParamStore(ecfg, gpu, torch.bfloat16, params_np=params)
    plan = E.StepPlan(store, 4, 128, lr=1e-2)
    plan.load_batch(batch["x"], batch["seq_lens"], batch["classes"], batch["labels"], eps)
    gs = []
    for _ in range(2):
        plan.step_kernels(True)
        torch.cuda.synchronize()
        gs.append(store.g.cpu().numpy().copy())
    runs.append((gs, store.w.cpu().numpy().copy(), {n: (o_, int(np.prod(s_))) for n, (o_, s_) in store.layout.items()} if hasattr(store, "layout") else None))
    del plan, store
ref = runs[0]
for i, (gs, w, _) in enumerate(runs):
    print(i, "w diff", f"{(np.abs(w - ref[1]) > 2e-5).mean():.4f}", "g1 rel", f"{np.abs(gs[0] - ref[0][0]).max() / np.abs(ref[0][0]).max():.2e}",
          "g2 rel", f"{np.abs(gs[1] - ref[0][1]).max() / np.abs(ref[0][1]).max():.2e}", flush=True)
bad = [i for i, (gs, w, _) in enumerate(runs) if (np.abs(w - ref[1]) > 2e-5).mean() > 0.05]
good = [i for i in range(len(runs)) if i not in bad]
if len(bad) > len(good):
    bad, good = good, bad
print("bad runs", bad)
if bad:
    st = E.ParamStore(ecfg, gpu, torch.bfloat16, params_np=params)
    g_good, g_bad = runs[good[0]][0][1], runs[bad[0]][0][1]
    g_good2 = runs[good[1]][0][1] if len(good) > 1 else g_good
    for n, o_ in st.offsets.items():
        sz = int(np.prod(st.shapes[n]))
        a, b, c = g_good[o_:o_ + sz], g_bad[o_:o_ + sz], g_good2[o_:o_ + sz]
        rel = np.abs(a - b).max() / (np.abs(a).max() + 1e-30)
        rel_gg = np.abs(a - c).max() / (np.abs(a).max() + 1e-30)
        cos = float((a * b).sum() / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
        if rel > 1e-3:
            print(f"  {n:40s} step-2 gradient: max rel diff {rel:.3f} (good vs good {rel_gg:.1e}) cos {cos:.4f} norm ratio {np.linalg.norm(b) / (np.linalg.norm(a) + 1e-30):.3f}")
