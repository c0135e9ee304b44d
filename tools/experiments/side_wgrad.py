"""Experiment: the early weight gradients (decoder, output layer, top encoder layer, latent block) flushed on a SIDE stream while
the rest of the backward pass runs on the main one — separate captured graphs joined by events (branches inside one hipGraph are
replayed back to back on one queue). configs[1]."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from musicstyletransfer_amd import engine as E, ops as o

dev = torch.device("cuda", 0); torch.cuda.set_device(0)
c = bench.CONFIGS[1]
B, T, P = c["B"], c["T"], c["P"]
cfg = E.VAEConfig(e_dropout=0.2, d_dropout=0.2, **bench.model_dims(c))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
main, side = torch.cuda.Stream(), torch.cuda.Stream()


def build():
    with torch.cuda.stream(main):
        store = E.ParamStore(cfg, dev, torch.bfloat16, seed=1234)
        plan = E.StepPlan(store, B, T, lr=3e-4, clip_gradient=1.0, kl_weight=1.0, global_batch=B, internal_eps=True, seed=5)
        hb = bench.synthetic_batches(1, B, T, P, seed=5)[0]
        plan.bind_inputs(plan.pack_batch(hb["x"], hb["seq_lens"], hb["classes"], hb["labels"]).to(dev))
        plan.step_kernels(True)
        torch.cuda.synchronize()
    return store, plan


def timed(fn, n):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


# ---- one graph
store0, plan0 = build()
with torch.cuda.stream(main):
    plan0.capture(True)


def one():
    with torch.cuda.stream(main):
        plan0.graph.launch()


# ---- four graphs
store, plan = build()
scratch2 = torch.zeros_like(plan.wgrad_scratch)


def g1():
    plan._tick_adam = True
    plan.forward()
    plan.losses(with_grad=True, combine=False)
    plan.backward_early(flush=False)


def gs():
    keep = plan.wgrad_scratch
    plan.wgrad_scratch = scratch2
    plan._flush_grads()
    plan.wgrad_scratch = keep


with torch.cuda.stream(main):
    G1 = o.Graph().capture(g1)
with torch.cuda.stream(side):
    GS = o.Graph().capture(gs)
with torch.cuda.stream(main):
    G2 = o.Graph().capture(plan.backward_late)
    G3 = o.Graph().capture(plan.optimizer)
with torch.cuda.stream(main):
    GSm = o.Graph().capture(lambda: None) if False else None
e1, es = torch.cuda.Event(), torch.cuda.Event()


def split_same_stream():
    with torch.cuda.stream(main):
        G1.launch(); GS_main.launch(); G2.launch(); G3.launch()


def split_side():
    with torch.cuda.stream(main):
        G1.launch()
        e1.record(main)
    with torch.cuda.stream(side):
        side.wait_event(e1)
        GS.launch()
        es.record(side)
    with torch.cuda.stream(main):
        G2.launch()
        main.wait_event(es)
        G3.launch()


print(f"one graph                         : {timed(one, N):.4f} ms per step")
print(f"four graphs, early flush on a side stream: {timed(split_side, N):.4f} ms per step")
# the same four pieces on ONE stream (the cost of cutting the graph): the side graph re-captured on the main stream
store2, plan2 = build()
plan = plan2
scratch2 = torch.zeros_like(plan.wgrad_scratch)
with torch.cuda.stream(main):
    H1 = o.Graph().capture(g1)
    GS_main = o.Graph().capture(gs)
    H2 = o.Graph().capture(plan.backward_late)
    H3 = o.Graph().capture(plan.optimizer)


def split_same():
    with torch.cuda.stream(main):
        H1.launch(); GS_main.launch(); H2.launch(); H3.launch()


print(f"four graphs, one stream           : {timed(split_same, N):.4f} ms per step")
print(f"one graph again                   : {timed(one, N):.4f} ms per step")
for st in (store0, store, store2):
    m = st.read_metrics()
    print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in m.items()})
