"""Experiment: the configs[1] step launched eagerly (25 ctypes calls per step) against its captured graph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from musicstyletransfer_amd import engine as E

dev = torch.device("cuda", 0); torch.cuda.set_device(0)
c = bench.CONFIGS[1]
B, T, P = c["B"], c["T"], c["P"]
cfg = E.VAEConfig(e_dropout=0.2, d_dropout=0.2, **bench.model_dims(c))
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    store = E.ParamStore(cfg, dev, torch.bfloat16, seed=1234)
    plan = E.StepPlan(store, B, T, lr=3e-4, clip_gradient=1.0, kl_weight=1.0, global_batch=B, internal_eps=True, seed=5)
    hb = bench.synthetic_batches(1, B, T, P, seed=5)[0]
    plan.bind_inputs(plan.pack_batch(hb["x"], hb["seq_lens"], hb["classes"], hb["labels"]).to(dev))
    for _ in range(5): plan.step_kernels(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100): plan.step_kernels(True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"eager: host loop {(t1 - t0) * 10:.4f} ms per step, with the final sync {(t2 - t0) * 10:.4f} ms per step")
    plan.capture(True); torch.cuda.synchronize()
    for _ in range(20): plan.graph.launch()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): plan.graph.launch()
    torch.cuda.synchronize()
    print(f"graph: {(time.perf_counter() - t0) / 300 * 1e3:.4f} ms per step")
