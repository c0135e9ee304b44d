"""Experiment (not product): do two half-batch steps replayed on two streams overlap on the card?
configs[1] widths; one B = 64 step against two concurrent B = 32 steps (each with its own store, so each half also pays
its own weight-gradient launch and optimizer — an upper bound on the cost of a split step)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from musicstyletransfer_amd import engine as E, ops as o

dev = torch.device("cuda", 0); torch.cuda.set_device(0)
c = bench.CONFIGS[1]
T, P = c["T"], c["P"]
md = bench.model_dims(c)
cfg = E.VAEConfig(e_dropout=0.2, d_dropout=0.2, **md)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300


def make(B, seed, stream):
    with torch.cuda.stream(stream):
        store = E.ParamStore(cfg, dev, torch.bfloat16, seed=1234)
        plan = E.StepPlan(store, B, T, lr=3e-4, clip_gradient=1.0, kl_weight=1.0, global_batch=B, internal_eps=True, seed=seed)
        hb = bench.synthetic_batches(1, B, T, P, seed=seed)[0]
        blob = plan.pack_batch(hb["x"], hb["seq_lens"], hb["classes"], hb["labels"]).to(dev)
        plan.bind_inputs(blob)
        plan.step_kernels(True)
        torch.cuda.synchronize()
        plan.capture(True)
        torch.cuda.synchronize()
    return store, plan, blob


def timed(fn, n):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
full = make(64, 1, sA)
hA = make(32, 2, sA)
hB = make(32, 3, sB)


def one_full():
    with torch.cuda.stream(sA):
        full[1].graph.launch()


def one_half():
    with torch.cuda.stream(sA):
        hA[1].graph.launch()


def two_halves():
    with torch.cuda.stream(sA):
        hA[1].graph.launch()
    with torch.cuda.stream(sB):
        hB[1].graph.launch()


def two_halves_serial():
    with torch.cuda.stream(sA):
        hA[1].graph.launch()
        hB[1].graph.launch()


print(f"B 64, one stream        : {timed(one_full, N):.4f} ms per step")
print(f"B 32, one stream        : {timed(one_half, N):.4f} ms per step")
print(f"2 x B 32, one stream    : {timed(two_halves_serial, N):.4f} ms per pair")
print(f"2 x B 32, two streams   : {timed(two_halves, N):.4f} ms per pair")
print(f"B 64 again              : {timed(one_full, N):.4f} ms per step")
for st in (full[0], hA[0], hB[0]):
    m = st.read_metrics()
    print("skipped steps:", m.get("skipped"), "count", m.get("count"))
