"""Experiment (not a bench line): the captured configs[1] step replayed at another batch size, for rocprofv3 --kernel-trace:
which launches of the step do not shrink with the batch (the fixed part of the step). usage: batch_sweep.py B [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from musicstyletransfer_amd import engine as E

dev = torch.device("cuda", 0); torch.cuda.set_device(0)
c = bench.CONFIGS[1]
T, P = c["T"], c["P"]
B = int(sys.argv[1]); N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
cfg = E.VAEConfig(e_dropout=0.2, d_dropout=0.2, **bench.model_dims(c))
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    store = E.ParamStore(cfg, dev, torch.bfloat16, seed=1234)
    plan = E.StepPlan(store, B, T, lr=3e-4, clip_gradient=1.0, kl_weight=1.0, global_batch=B, internal_eps=True, seed=7)
    hb = bench.synthetic_batches(1, B, T, P, seed=7)[0]
    plan.bind_inputs(plan.pack_batch(hb["x"], hb["seq_lens"], hb["classes"], hb["labels"]).to(dev))
    plan.step_kernels(True); torch.cuda.synchronize()
    plan.capture(True); torch.cuda.synchronize()
    for _ in range(20): plan.graph.launch()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(N): plan.graph.launch()
    torch.cuda.synchronize()
    print(f"B {B}: {(time.perf_counter() - t0) / N * 1e3:.4f} ms per step")
