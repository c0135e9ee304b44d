"""Data parallelism for the VAE step: one process per GPU, the minibatch sharded contiguously, ONE
collective per step — a SUM all-reduce of the flat fp32 gradient bucket (RCCL over xGMI; `nccl` IS RCCL
on ROCm) — followed by the same fused Adam on every rank.

The reference has no counterpart (single context everywhere: main.py:124, trainer.py:75,99-101); the
only cross-sample operation in its step is the implicit sum over the batch in loss.backward() plus the
1/B of Trainer.step(batch_size) (trainer.py:176-177). Hence:
  * every rank normalises by the GLOBAL batch size (StepPlan(global_batch=...)),
  * every rank pads to the GLOBAL batch-max length, because SoftmaxCrossEntropy divides by the padded
    length T (loss.py:23) and data.py:196-198 truncates to the batch maximum,
  * parameters, Adam state and positional tables are replicated; initial weights are identical on every
    rank (same seed), and identical reduced gradients keep them identical.
The bucket is 7.5 MB (configs[1]); xGMI is point-to-point (7 links x ~153 GB/s per GPU), so the
collective costs tens of microseconds and is issued as a single call on the step's stream between the
captured forward/backward graph and the captured optimizer graph.
"""
import glob
import os
import re
import tempfile
import time

import numpy as np
import torch


def init_process_group(world, rank, backend=None):
    """env:// rendezvous on 127.0.0.1 (the container hostname may not resolve). The launcher (torch.distributed.run)
    exports MASTER_PORT; the fallback below only serves hand-started ranks, which must agree on a port anyway.
    Call torch.cuda.set_device() for this rank's GPU FIRST: the nccl (= RCCL) group binds to the current device.
    MST_DIST_BACKEND=gloo rehearses the multi-rank path with the ranks sharing one card (tests, one-GPU boxes)."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this host driver
    if backend is None:
        backend = os.environ.get("MST_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        rccl_debug_env(rank)
    if not dist.is_initialized():
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", torch.cuda.current_device())
        dist.init_process_group(backend=backend, init_method="env://", world_size=world, rank=rank, **kw)
    return dist


def make_grad_allreduce(dist, group=None):
    """returns reduce_fn(flat_grad): in-place SUM over ranks, asynchronous on the current stream (nccl) or
    blocking (gloo, CPU tests)"""
    def reduce_fn(flat):
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return reduce_fn


# ------------------------------------------------------------------------------------------------ RCCL: what it does, and which algorithm
# RCCL keeps NCCL's environment interface. NCCL_DEBUG=INFO with the INIT and TUNING subsystems makes every communicator
# describe itself when it is created (ranks, channels, transports) and every collective report the algorithm / protocol
# the tuner chose for its size ("AllReduce: 3302912 Bytes -> Algo 1 proto 2 ..."). The lines go to a per-process file
# (NCCL_DEBUG_FILE), never to stdout — rank 0's stdout is bench.py's ONE JSON line.
_ALGOS = {"0": "Tree", "1": "Ring", "2": "CollNetDirect", "3": "CollNetChain", "4": "NVLS", "5": "NVLSTree", "6": "PAT"}
_ALGO_NAMES = {v.upper(): v for v in _ALGOS.values()}
_PROTOS = {"0": "LL", "1": "LL128", "2": "Simple", "SIMPLE": "Simple"}


def rccl_debug_env(rank):
    """OPT-IN (MST_RCCL_DEBUG=1; bench.py sets it for its own runs, a training run leaves RCCL's environment alone): ask
    RCCL, before its first communicator exists, to log its SET-UP (subsystems INIT and ENV: version, channels, transports
    per channel, the NCCL_* variables it read) to a file of this process. Nothing is logged per collective, so nothing
    lands inside a timed loop or grows with the run. MST_RCCL_DEBUG=tuning adds the TUNING subsystem (one line per
    all-reduce with the algorithm / protocol chosen: for a short diagnostic run, not for timing). An NCCL_DEBUG the
    caller exported wins. The log directory is MST_RCCL_LOG_DIR or a fresh temporary one, which rccl_report(cleanup=True)
    removes again."""
    mode = os.environ.get("MST_RCCL_DEBUG", "0")
    if mode in ("0", "") or "NCCL_DEBUG" in os.environ:
        return None
    d = os.environ.get("MST_RCCL_LOG_DIR")
    if not d:
        d = tempfile.mkdtemp(prefix="mst_rccl_")
        os.environ["MST_RCCL_LOG_TMP"] = d  # ours to remove
    os.makedirs(d, exist_ok=True)
    os.environ["NCCL_DEBUG"] = "INFO"
    os.environ["NCCL_DEBUG_SUBSYS"] = "INIT,ENV,TUNING" if mode == "tuning" else "INIT,ENV"
    os.environ["NCCL_DEBUG_FILE"] = os.path.join(d, f"rank{rank}.%p.log")
    os.environ["MST_RCCL_LOG_DIR"] = d
    return d


def parse_rccl_log(text):
    """What an NCCL_DEBUG=INFO log says about the communicators and the all-reduces of this process:
    {'version', 'nranks', 'channels', 'transports': [...], 'allreduce': [{'bytes', 'algo', 'proto', 'calls'}, ...],
     'env': {NCCL_* / RCCL_* variables RCCL says it read}}. Unknown formats leave fields out rather than guessing."""
    out = {}
    m = re.search(r"(?:RCCL|NCCL) version[ :]+([^\s]+)", text)
    if m:
        out["version"] = m.group(1)
    m = re.findall(r"nranks (\d+)", text)
    if m:
        out["nranks"] = max(int(v) for v in m)
    m = re.findall(r"(\d+) coll channels", text)
    if m:
        out["channels"] = max(int(v) for v in m)
    tr = sorted(set(re.findall(r"\bvia ([A-Za-z0-9/_]+)", text)))
    if tr:
        out["transports"] = tr
    calls = {}
    for size, algo, proto in re.findall(r"AllReduce: (\d+) Bytes -> Algo (\S+) proto (\S+)", text):
        key = (int(size), _ALGOS.get(algo, _ALGO_NAMES.get(algo.upper(), algo)), _PROTOS.get(proto, proto.upper()))
        calls[key] = calls.get(key, 0) + 1
    if calls:
        out["allreduce"] = [dict(bytes=k[0], algo=k[1], proto=k[2], calls=n) for k, n in sorted(calls.items())]
    env = dict(re.findall(r"\b((?:NCCL|RCCL)_[A-Z0-9_]+) set by environment to (\S+)", text))
    if env:
        out["env"] = env
    return out


def rccl_report(cleanup=False):
    """parse this process's RCCL log(s) (rccl_debug_env); {} when logging is off or nothing was written. cleanup: remove
    this rank's log files, and the directory if rccl_debug_env created it and it is empty."""
    d = os.environ.get("MST_RCCL_LOG_DIR")
    if not d:
        return {}
    text = ""
    files = sorted(glob.glob(os.path.join(d, f"rank{os.environ.get('RANK', '0')}.*log*")))
    for f in files:
        try:
            with open(f, errors="replace") as fh:
                text += fh.read()
        except OSError:
            pass
    if cleanup and os.environ.get("MST_RCCL_LOG_TMP") == d:
        for f in files:
            try:
                os.remove(f)
            except OSError:
                pass
        try:
            os.rmdir(d)  # (the last rank to leave removes it; fails harmlessly while other ranks' files are there)
        except OSError:
            pass
    return parse_rccl_log(text) if text else {}


def _time_allreduce(dist, group, tensors, iters, sync):
    """average microseconds of one round of SUM all-reduces over `tensors` on `group`, wall clock around `iters` rounds
    bracketed by sync() (a device synchronize for nccl), MAX over ranks so that every rank sees the same number"""
    sync()
    dist.barrier(group=group)
    sync()
    t0 = time.perf_counter()
    for _ in range(iters):
        for t in tensors:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    sync()
    us = torch.tensor([(time.perf_counter() - t0) / iters * 1e6], dtype=torch.float64, device=tensors[0].device)
    dist.all_reduce(us, op=dist.ReduceOp.MAX, group=group)
    return float(us.item())


def autotune_allreduce(dist, range_numels, device, candidates=None, iters=10, rounds=5, margin=0.07):
    """OPT-IN (MST_RCCL_AUTOTUNE=1, or MST_RCCL_ALGO=<name> to pin without measuring; off by default until it has been
    measured once on an 8-GPU node). Measure, on this job's own ranks and links, the all-reduce of the step's gradient
    ranges under each candidate RCCL algorithm and return (group, report): the process group whose communicator was
    created under the chosen setting (None = the default group) and {'candidates': {name: median microseconds per
    step's worth of all-reduces}, 'samples': {...}, 'chosen': name}.

    RCCL reads NCCL_ALGO when a communicator is created, so each candidate is a dist.new_group() made under that
    environment value; "default" leaves the choice to RCCL's tuner (per message size). The candidates are timed in
    INTERLEAVED rounds (default, Tree, Ring, default, ...) after a warm-up of each, and compared by their MEDIANS: one
    sequential pass per candidate measured order and warm-up instead (profiles/r03_dp2_rehearsal_gloo_v2.json: 5240 / 3619 /
    3502 us for three functionally identical gloo groups). The default communicator is kept unless a candidate beats it
    by `margin` (7 %): a pinned algorithm overrides the tuner for every message size, so it has to earn that. Every rank
    runs the same sequence and every timing is MAX-reduced, so all ranks choose alike."""
    backend = dist.get_backend()
    sync = torch.cuda.synchronize if backend == "nccl" else (lambda: None)
    pinned = os.environ.get("MST_RCCL_ALGO")
    if candidates is None:
        candidates = [pinned] if pinned else ["default", "Tree", "Ring"]
    report = {"candidates": {}, "chosen": "default", "range_bytes": [4 * n for n in range_numels], "margin": margin}
    if os.environ.get("MST_RCCL_AUTOTUNE", "0") != "1" and not pinned:
        report["skipped"] = "MST_RCCL_AUTOTUNE is not 1"
        return None, report
    bufs = [torch.zeros(n, dtype=torch.float32, device=device) for n in range_numels if n > 0]
    groups, usable = {}, []
    saved = os.environ.get("NCCL_ALGO")
    try:
        for name in candidates:
            try:
                if name == "default":
                    os.environ.pop("NCCL_ALGO", None)
                    groups[name] = None if saved is None else dist.new_group()
                else:
                    os.environ["NCCL_ALGO"] = name
                    groups[name] = dist.new_group()
                for t in bufs:  # warm-up: lazy communicator creation, first-call set-up
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=groups[name])
                sync()
                usable.append(name)
            except RuntimeError as e:  # an algorithm this RCCL build refuses for all-reduce: drop the candidate, on every rank alike
                report["candidates"][name] = None
                report.setdefault("errors", {})[name] = str(e).splitlines()[0][:200]
    finally:
        if saved is None:
            os.environ.pop("NCCL_ALGO", None)
        else:
            os.environ["NCCL_ALGO"] = saved
    if pinned:
        report["candidates"][pinned] = None
        report["chosen"] = pinned if pinned in usable else "default"
        return groups.get(report["chosen"]), report
    samples = {name: [] for name in usable}
    for _ in range(rounds):
        for name in usable:
            samples[name].append(_time_allreduce(dist, groups[name], bufs, iters, sync))
    report["samples"] = samples
    for name, v in samples.items():
        report["candidates"][name] = sorted(v)[len(v) // 2]
    timed = {k: v for k, v in report["candidates"].items() if v is not None}
    if timed:
        fastest = min(timed, key=timed.get)
        ref = timed.get("default")
        if ref is None or (fastest != "default" and timed[fastest] <= (1.0 - margin) * ref):
            report["chosen"] = fastest
        else:
            report["chosen"] = "default"
            if fastest != "default":
                report["kept_default"] = f"{fastest} was faster by less than the margin ({timed[fastest]:.1f} vs {ref:.1f} us)"
    return groups.get(report["chosen"]), report


class GradReducer:
    """Asynchronous SUM all-reduce of ranges of the flat gradient bucket. start() enqueues the collective behind
    everything already issued on the current stream (nccl: on RCCL's own stream, so kernels launched afterwards on
    the current stream overlap with it; gloo: a host-side work item), finish() makes the current stream (nccl) or the
    host (gloo) wait for the given handles. This is the comm/compute overlap of the data-parallel step: the part of
    the bucket that the top of the backward pass completes travels over xGMI while the rest of backward executes.
    group: the process group (= RCCL communicator) the collectives run on — the one autotune_allreduce() picked, or
    the default group."""

    def __init__(self, dist, group=None):
        self.dist, self.group = dist, group

    def start(self, flat_range):
        return self.dist.all_reduce(flat_range, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self, handles):
        for h in handles:
            h.wait()


def shard_bounds(global_batch, world, rank):
    """contiguous shard [lo, hi) of the global batch for `rank`"""
    assert global_batch % world == 0, "global batch must divide evenly over the ranks"
    per = global_batch // world
    return rank * per, (rank + 1) * per


def shard_batch(batch, world, rank):
    """slice every per-sample array of a batch dict; the padded length (axis 1) is left at the global
    batch maximum on purpose (see module docstring)"""
    B = len(batch["seq_lens"])
    lo, hi = shard_bounds(B, world, rank)
    return {k: v[lo:hi] for k, v in batch.items()}


def global_eps(seed, step, global_batch, latent_dim, lo, hi):
    """Host-side eps drawn per GLOBAL sample index, for callers that INJECT eps (StepPlan(internal_eps=False)): results
    then do not depend on how the batch is sharded. The in-graph draw of the training step follows the same rule on the
    device: mst_step_begin(eps_index0 = first global sample index * Z), see StepPlan(sample_offset=...)."""
    rng = np.random.default_rng([seed, step])
    return rng.standard_normal((global_batch, latent_dim)).astype(np.float32)[lo:hi]
