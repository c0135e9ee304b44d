"""bench.py started as `python bench.py --gpus N` with no launcher around it must start its own N ranks before any GPU
call, hand them torch.distributed.run's environment, relay rank 0's one JSON line and fail when a rank fails
(VERDICT r02 item 1). Tested here without a GPU through --dry-launch (the ranks print their environment and exit), plus
the pieces of parallel.py that report on / choose RCCL's all-reduce, with gloo standing in at world size 2."""
import json
import os
import socket
import subprocess
import sys
import time

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(extra)
    return env


def test_self_launch_starts_one_rank_per_gpu_with_the_rendezvous_environment():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--steps", "3", "--dry-launch"], env=_env(), capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    out = [json.loads(l) for l in r.stdout.splitlines() if l.strip()]
    assert len(out) == 1 and out[0]["RANK"] == "0"  # stdout carries rank 0's line only
    others = [json.loads(l) for l in r.stderr.splitlines() if l.startswith("{")]
    ranks = out + others
    assert sorted(int(e["RANK"]) for e in ranks) == [0, 1, 2, 3]
    assert len({e["MASTER_PORT"] for e in ranks}) == 1 and int(ranks[0]["MASTER_PORT"]) > 0
    for e in ranks:
        assert e["WORLD_SIZE"] == e["LOCAL_WORLD_SIZE"] == "4" and e["LOCAL_RANK"] == e["RANK"]
        assert e["MASTER_ADDR"] == "127.0.0.1" and e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_self_launch_is_skipped_under_an_outside_launcher():
    """with WORLD_SIZE in the environment (torch.distributed.run) the process IS a rank: nothing is spawned"""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-launch"], capture_output=True, text=True, timeout=60,
                       env=_env(WORLD_SIZE="2", RANK="1", LOCAL_RANK="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="1"))
    assert r.returncode == 0 and json.loads(r.stdout)["RANK"] == "1" and json.loads(r.stdout)["MST_BENCH_SELF_LAUNCHED"] is None


def test_a_failing_rank_fails_the_launcher_and_stops_the_others():
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "3", "--dry-launch"], capture_output=True, text=True, timeout=120,
                       env=_env(MST_BENCH_DRY_FAIL_RANK="2", MST_BENCH_DRY_SLEEP="60"))
    assert r.returncode == 3 and r.stdout == ""
    assert "rank 2 exited with 3" in r.stderr
    assert time.monotonic() - t0 < 30  # the sleeping ranks were terminated, not waited for


def test_parse_rccl_log():
    from musicstyletransfer_amd import parallel as P
    text = """h:1:2 [0] NCCL INFO NCCL version 2.22.3+hip7.0
h:1:2 [0] NCCL INFO NCCL_ALGO set by environment to Tree
h:1:2 [0] NCCL INFO comm 0x1 rank 0 nranks 8 cudaDev 0 busId 1000 commId 0x1 - Init START
h:1:2 [0] NCCL INFO Channel 00/0 : 0[0] -> 1[1] via P2P/IPC
h:1:2 [0] NCCL INFO 32 coll channels, 32 collnet channels, 0 nvls channels, 32 p2p channels
h:1:2 [0] NCCL INFO AllReduce: 3302912 Bytes -> Algo 1 proto 2 time 55.3
h:1:2 [0] NCCL INFO AllReduce: 3302912 Bytes -> Algo 1 proto 2 time 55.3
h:1:2 [0] NCCL INFO AllReduce: 4238848 Bytes -> Algo TREE proto LL128 channel{Lo..Hi}={0..15}
"""
    rep = P.parse_rccl_log(text)
    assert rep["version"].startswith("2.22") and rep["nranks"] == 8 and rep["channels"] == 32 and rep["transports"] == ["P2P/IPC"]
    assert rep["allreduce"] == [dict(bytes=3302912, algo="Ring", proto="Simple", calls=2), dict(bytes=4238848, algo="Tree", proto="LL128", calls=1)]
    assert rep["env"] == {"NCCL_ALGO": "Tree"}
    assert P.parse_rccl_log("nothing of interest") == {}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _tune_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    from musicstyletransfer_amd import parallel
    dist = parallel.init_process_group(world, rank, backend="gloo")
    _, off = parallel.autotune_allreduce(dist, [1000, 600], torch.device("cpu"), iters=3)  # opt-in: nothing is measured by default
    assert off["chosen"] == "default" and off["candidates"] == {} and "skipped" in off
    os.environ["MST_RCCL_AUTOTUNE"] = "1"
    group, rep = parallel.autotune_allreduce(dist, [1000, 600], torch.device("cpu"), iters=3, rounds=3)
    flat = torch.full((1600,), float(rank + 1))
    red = parallel.GradReducer(dist, group)
    red.finish([red.start(flat[600:]), red.start(flat[:600])])
    whole = torch.full((1600,), float(rank + 1))
    parallel.make_grad_allreduce(dist, group)(whole)
    os.environ["MST_RCCL_ALGO"] = "Tree"  # pinned: no measurement, a communicator made under that setting
    g2, rep2 = parallel.autotune_allreduce(dist, [1000, 600], torch.device("cpu"))
    q.put((rank, rep, bool(torch.all(flat == 3.0)) and bool(torch.all(whole == 3.0)), rep2, g2 is not None, os.environ.get("NCCL_ALGO")))
    dist.barrier()
    dist.destroy_process_group()


def test_autotune_picks_one_group_on_every_rank_and_the_reducer_uses_it():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tune_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, rep0, ok0, pin0, g0, env0), (_, rep1, ok1, pin1, g1, env1) = res
    assert ok0 and ok1
    assert rep0["chosen"] == rep1["chosen"] and rep0["candidates"] == rep1["candidates"]  # MAX-reduced timings: same choice everywhere
    assert set(rep0["candidates"]) == {"default", "Tree", "Ring"} and all(v > 0 for v in rep0["candidates"].values())
    assert rep0["range_bytes"] == [4000, 2400]
    assert all(len(v) == 3 for v in rep0["samples"].values())  # interleaved rounds, compared by their medians
    if rep0["chosen"] != "default":  # the default communicator is kept unless a candidate beats it by the margin
        assert rep0["candidates"][rep0["chosen"]] <= (1 - rep0["margin"]) * rep0["candidates"]["default"]
    assert pin0["chosen"] == "Tree" and pin0["candidates"] == {"Tree": None} and g0 and g1
    assert env0 is None and env1 is None  # the caller's NCCL_ALGO is restored


# ---------------------------------------------------------------------------------- the N > 1 bench line cannot come back empty
def _bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", BENCH)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_a_raising_candidate_leaves_the_baseline_number_intact_and_ends_the_experiments():
    b = _bench()
    base = {"ms_per_step": 1.0, "tag": "baseline"}
    ran = []

    def boom():
        ran.append("boom")
        raise RuntimeError("NCCL error: unhandled system error\nsecond line")

    def never():
        ran.append("never")
        return {"ms_per_step": 0.1}
    name, res, rep = b.run_candidates("plain", base, [("overlap", boom), ("tuned", never)], budget_s=100.0)
    assert (name, res) == ("plain", base) and ran == ["boom"]  # after a failed collective nothing else is tried
    assert rep["chosen"] == "plain" and rep["baseline_ms"] == 1.0 and rep["candidates"] == {"overlap": None}
    assert rep["errors"]["overlap"].startswith("RuntimeError: NCCL error") and rep["not_run"] == ["tuned"]


def test_candidates_replace_the_baseline_only_when_faster_and_within_the_budget():
    b = _bench()
    base = {"ms_per_step": 1.0}
    clock = iter([0.0, 1.0, 2.0, 50.0, 51.0, 52.0]).__next__
    seen = []
    cands = [("slower", lambda: {"ms_per_step": 1.2}), ("faster", lambda: {"ms_per_step": 0.8}),
             ("rejected", lambda: (_ for _ in ()).throw(b.CandidateRejected("warm-up 9 ms > 3 ms"))),
             ("too late", lambda: {"ms_per_step": 0.1})]
    name, res, rep = b.run_candidates("plain", base, cands, budget_s=10.0, clock=clock, on_best=lambda n, r: seen.append(n))
    assert name == "faster" and res["ms_per_step"] == 0.8 and seen == ["faster"]
    assert rep["candidates"] == {"slower": 1.2, "faster": 0.8}
    assert rep["not_run"] == ["rejected", "too late"] and rep["chosen"] == "faster"
    # a rejection (decided alike on every rank) does not end the experiments
    name, res, rep = b.run_candidates("plain", base, cands[2:], budget_s=1e9)
    assert name == "too late" and rep["errors"]["rejected"].startswith("rejected:") and rep["candidates"]["rejected"] is None


def test_rccl_logging_is_opt_in_and_cleans_up(tmp_path, monkeypatch):
    from musicstyletransfer_amd import parallel as P
    for k in ("NCCL_DEBUG", "NCCL_DEBUG_SUBSYS", "NCCL_DEBUG_FILE", "MST_RCCL_LOG_DIR", "MST_RCCL_LOG_TMP", "MST_RCCL_DEBUG"):
        monkeypatch.delenv(k, raising=False)
    assert P.rccl_debug_env(0) is None and "NCCL_DEBUG" not in os.environ  # a training run leaves RCCL's environment alone
    monkeypatch.setenv("MST_RCCL_DEBUG", "1")
    monkeypatch.setenv("RANK", "0")
    d = P.rccl_debug_env(0)
    try:
        assert os.environ["NCCL_DEBUG_SUBSYS"] == "INIT,ENV"  # nothing per collective: no TUNING lines inside a timed loop
        with open(os.path.join(d, "rank0.123.log"), "w") as fh:
            fh.write("NCCL INFO NCCL version 2.22.3+hip7.0\n")
        assert P.rccl_report(cleanup=True)["version"].startswith("2.22")
        assert not os.path.exists(d)
    finally:
        for k in ("NCCL_DEBUG", "NCCL_DEBUG_SUBSYS", "NCCL_DEBUG_FILE", "MST_RCCL_LOG_DIR", "MST_RCCL_LOG_TMP"):
            os.environ.pop(k, None)
