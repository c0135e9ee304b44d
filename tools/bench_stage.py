"""Where does the host side of PinnedBatchPipeline.stage() spend its time? (diagnostic, GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from musicstyletransfer_amd import engine as E
from musicstyletransfer_amd.pianoroll import PinnedBatchPipeline, _Slot
from musicstyletransfer_amd.VarAutoEncoder.data import Batch
import bench

c = bench.CONFIGS[1]
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
cfg = E.VAEConfig(**bench.model_dims(c))
store = E.ParamStore(cfg, dev, torch.bfloat16)
plan = E.StepPlan(store, c["B"], c["T"])
hb = bench.synthetic_batches(2, c["B"], c["T"], c["P"], 1)
slot = _Slot(plan.own_inbuf.numel(), dev)
def t(fn, n=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
b = hb[0]
print("pack_into pinned      : %.3f ms" % t(lambda: plan.pack_into(slot.host, b["x"], b["seq_lens"], b["classes"], b["labels"])))
pageable = torch.zeros(plan.own_inbuf.numel(), dtype=torch.uint8)
print("pack_into pageable    : %.3f ms" % t(lambda: plan.pack_into(pageable, b["x"], b["seq_lens"], b["classes"], b["labels"])))
x = torch.from_numpy(b["x"]); dst = slot.host[: x.numel()].view(c["B"] * c["T"], c["P"])
print("raw copy_ 2 MB pinned : %.3f ms" % t(lambda: dst.copy_(x.reshape(c["B"] * c["T"], c["P"]))))
dn = slot.host.numpy()
print("numpy copy 2 MB pinned: %.3f ms" % t(lambda: np.copyto(dn[: x.numel()].reshape(b["x"].shape), b["x"])))
print("H2D 4 MB non_blocking : %.3f ms" % t(lambda: slot.dev.copy_(slot.host, non_blocking=True)))
pipe = PinnedBatchPipeline(dev, lambda B, T: plan)
bt = Batch([b["x"], b["seq_lens"], b["classes"]], [b["labels"]])
print("pipe.stage            : %.3f ms" % t(lambda: pipe.stage(bt)))
print("torch threads", torch.get_num_threads())

# ---- the host-fed step loop of bench.py --data host, timed part by part
from musicstyletransfer_amd import ops as o
cfg2 = E.VAEConfig(e_dropout=0.2, d_dropout=0.2, **bench.model_dims(c))
store2 = E.ParamStore(cfg2, dev, torch.bfloat16)
plan2 = E.StepPlan(store2, c["B"], c["T"], internal_eps=True)
pipe2 = PinnedBatchPipeline(dev, lambda B, T: plan2)
host = bench.synthetic_batches(4, c["B"], c["T"], c["P"], 7)
batches = [Batch([h["x"], h["seq_lens"], h["classes"]], [h["labels"]]) for h in host]
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    st0 = [pipe2.stage(batches[i % 4]) for i in range(3)]
    plan2.bind_inputs(st0[0].slot.dev)
    plan2.step_kernels(True)
    torch.cuda.synchronize()
    graphs = {}
    for s in st0:
        plan2.bind_inputs(s.slot.dev)
        plan2.capture(True)
        graphs[s.slot.dev.data_ptr()] = plan2.graph
    N = 300
    feed = pipe2.feed(batches[i % 4] for i in range(N))
    rows = []
    torch.cuda.synchronize()
    T0 = time.perf_counter()
    for i in range(N):
        t0 = time.perf_counter(); s = next(feed)
        t1 = time.perf_counter(); stream.wait_event(s.slot.uploaded)
        t2 = time.perf_counter(); graphs[s.slot.dev.data_ptr()].launch()
        t3 = time.perf_counter(); s.slot.consumed.record(stream)
        t4 = time.perf_counter()
        rows.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
    torch.cuda.synchronize()
    tot = time.perf_counter() - T0
r = np.array(rows) * 1e3
print("host-fed loop: %.3f ms/step wall; per part (ms) median / max:" % (tot / N * 1e3))
for k, name in enumerate(("next(feed)=stage(i+1)", "wait_event", "graph launch", "record")):
    print("   %-24s %.3f / %.3f   (sum %.1f ms)" % (name, np.median(r[:, k]), r[:, k].max(), r[:, k].sum()))
slow = np.argsort(-r.sum(1))[:8]
print("   slowest steps:", [(int(i), [round(float(v), 2) for v in r[i]]) for i in sorted(slow)])
