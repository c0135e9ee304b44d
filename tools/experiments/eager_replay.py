"""Experiment: the step's launches recorded once (bound ctypes functions + their argument objects) and re-issued from a tight
Python loop — no graph — on one stream, and with the early weight-gradient flush on a side stream (events between the two)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from musicstyletransfer_amd import engine as E, ops as o, _lib

dev = torch.device("cuda", 0); torch.cuda.set_device(0)
c = bench.CONFIGS[1]
B, T, P = c["B"], c["T"], c["P"]
cfg = E.VAEConfig(e_dropout=0.2, d_dropout=0.2, **bench.model_dims(c))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
main, side = torch.cuda.Stream(), torch.cuda.Stream()
lib = _lib.load()
rec = None
orig_call = o.call


def rec_call(name, *args):
    if rec is not None:
        rec.append((getattr(lib, name), args, name))
    return orig_call(name, *args)


o.call = rec_call


def record(fn):
    global rec
    rec = []
    fn()
    out, rec = rec, None
    return out


def replay(calls, stream_from=None, stream_to=None):
    for f, a, _ in calls:
        if stream_to is not None:
            a = tuple(stream_to if (isinstance(x, int) and x == stream_from) else x for x in a)
        f(*a)


def timed(fn, n):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, (t1 - t0) / n * 1e3


with torch.cuda.stream(main):
    store = E.ParamStore(cfg, dev, torch.bfloat16, seed=1234)
    plan = E.StepPlan(store, B, T, lr=3e-4, clip_gradient=1.0, kl_weight=1.0, global_batch=B, internal_eps=True, seed=5)
    hb = bench.synthetic_batches(1, B, T, P, seed=5)[0]
    plan.bind_inputs(plan.pack_batch(hb["x"], hb["seq_lens"], hb["classes"], hb["labels"]).to(dev))
    plan.step_kernels(True); torch.cuda.synchronize()
    whole = record(lambda: plan.step_kernels(True))
    print(len(whole), "calls per step:", " ".join(n for _, _, n in whole))
    scratch2 = torch.zeros_like(plan.wgrad_scratch)

    def g1():
        plan._tick_adam = True
        plan.forward(); plan.losses(with_grad=True, combine=False); plan.backward_early(flush=False)

    def gs():
        keep = plan.wgrad_scratch
        plan.wgrad_scratch = scratch2
        plan._flush_grads()
        plan.wgrad_scratch = keep

    L1 = record(g1); LS = record(gs); L2 = record(plan.backward_late); L3 = record(plan.optimizer)
    torch.cuda.synchronize()
    plan.capture(True); torch.cuda.synchronize()
ms, ss = main.cuda_stream, side.cuda_stream
e1, es = torch.cuda.Event(), torch.cuda.Event()


def graph():
    with torch.cuda.stream(main):
        plan.graph.launch()


def eager_one():
    replay(whole)


def eager_split_same():
    replay(L1); replay(LS); replay(L2); replay(L3)


def eager_side():
    replay(L1)
    e1.record(main)
    side.wait_event(e1)
    replay(LS, ms, ss)
    es.record(side)
    replay(L2)
    main.wait_event(es)
    replay(L3)


for name, fn in (("graph", graph), ("recorded launches, one stream", eager_one), ("recorded, early flush apart, one stream", eager_split_same),
                 ("recorded, early flush on a side stream", eager_side), ("graph", graph)):
    t, h = timed(fn, N)
    print(f"{name:42s}: {t:.4f} ms per step (host loop {h:.4f})")
print(store.read_metrics())
