import os
import socket
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DP_WORLD = 2
_dp = {"procs": None, "dir": None}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def _gpu_run_selected(config):
    expr = (config.getoption("-m") or "").strip()
    return "gpu" in expr and "not gpu" not in expr


def pytest_sessionstart(session):
    """The data-parallel GPU tests need ranks that are their own processes. They are started HERE, before this process
    has made any HIP call (a process that has initialised the GPU must not start other programs on this pool), run
    beside the rest of the suite on the same card, and are collected by tests/test_parallel_gpu.py."""
    if not _gpu_run_selected(session.config):
        return
    try:
        import torch
        if torch.cuda.device_count() < 1:  # (counting devices does not initialise HIP)
            return
    except Exception:
        return
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = tempfile.mkdtemp(prefix="mst_dp_")
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    procs = []
    for r in range(DP_WORLD):
        log = open(os.path.join(out, f"rank{r}.log"), "w")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), "--rank", str(r), "--world",
                                       str(DP_WORLD), "--port", str(port), "--out", out], stdout=log, stderr=subprocess.STDOUT, env=env))
    _dp["procs"], _dp["dir"] = procs, out


def pytest_sessionfinish(session, exitstatus):
    for p in _dp["procs"] or []:
        if p.poll() is None:
            p.kill()  # exactly the processes started above


@pytest.fixture(scope="session")
def dp_results():
    """outputs of the DP_WORLD worker ranks (tests/dp_worker.py): list of dicts, rank order"""
    import numpy as np
    if _dp["procs"] is None:
        pytest.skip("data-parallel workers were not started (run with -m gpu on a GPU box)")
    for r, p in enumerate(_dp["procs"]):
        try:
            p.wait(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()
            pytest.fail(f"data-parallel worker rank {r} did not finish; log:\n" + open(os.path.join(_dp['dir'], f'rank{r}.log')).read()[-4000:])
    res = []
    for r, p in enumerate(_dp["procs"]):
        f = os.path.join(_dp["dir"], f"rank{r}.npz")
        if p.returncode != 0 or not os.path.exists(f):
            err = os.path.join(_dp["dir"], f"rank{r}.err")
            msg = open(err).read() if os.path.exists(err) else open(os.path.join(_dp["dir"], f"rank{r}.log")).read()[-4000:]
            pytest.fail(f"data-parallel worker rank {r} failed (exit {p.returncode}):\n{msg}")
        with np.load(f) as z:
            res.append({k: z[k] for k in z.files})
    return res


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from musicstyletransfer_amd import _lib
    _lib.load()
    return torch.device("cuda", 0)
