"""When does each workgroup of the step's weight-gradient launch start and end? Build with
MST_EXTRA_FLAGS="gemm_wgrad.hip=-DMST_WGRAD_STAMPS"; runs configs[1] steps eagerly and reads the per-workgroup realtime stamps
of the last launch (100 MHz), grouped by problem."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from musicstyletransfer_amd import engine as E, ops as o, _lib
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
c = bench.CONFIGS[1]
cfg = E.VAEConfig(e_dropout=0.2, d_dropout=0.2, **bench.model_dims(c))
store = E.ParamStore(cfg, dev, torch.bfloat16, seed=1234)
plan = E.StepPlan(store, c["B"], c["T"], lr=3e-4, clip_gradient=1.0, internal_eps=True, seed=1000)
hb = bench.synthetic_batches(1, c["B"], c["T"], c["P"], seed=1)[0]
plan.bind_inputs(plan.pack_batch(hb["x"], hb["seq_lens"], hb["classes"], hb["labels"]).to(dev))
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for _ in range(3):
        plan.step_kernels(True)
    torch.cuda.synchronize()
lib = _lib.load()
out = (C.c_uint64 * 1024)()
assert lib.mst_debug_wgrad_wg(out) == 0
t = np.array(list(out), dtype=np.int64).reshape(512, 2)
live = t[:, 1] > 0
t0 = t[live, 0].min()
n_items = int(live.sum())
print(f"{n_items} workgroups; launch span {(t[live, 1].max() - t0) / 100.0:.1f} us")
probs = plan._last_wgrads if hasattr(plan, "_last_wgrads") else None
start, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0
# items are numbered problem, slab, tile: print a histogram of end times in item order (groups of 8)
order = np.nonzero(live)[0]
for i in range(0, len(order), 16):
    idx = order[i:i + 16]
    print("items %3d-%3d: start %5.1f..%5.1f  end %5.1f..%5.1f  dur %5.1f..%5.1f" % (idx[0], idx[-1], start[idx].min(), start[idx].max(), end[idx].min(),
                                                                                   end[idx].max(), (end[idx] - start[idx]).min(), (end[idx] - start[idx]).max()))
