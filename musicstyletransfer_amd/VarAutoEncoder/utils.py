"""File helpers (reference VarAutoEncoder/utils.py:15-70): checkpoint discovery, parameter and training-state
persistence. On-disk names are the reference's (`params.N`, `train_state.pkl`); the parameter format is a
numpy .npz of name -> array (the reference's is MXNet's binary NDArray file). Fixed here: the checkpoint
index regex (`(\\d)+` there captures only the last digit, utils.py:18-20)."""
import os
import pickle
import re


class Context:
    """what mx.gpu() / mx.cpu() are in the reference (main.py:124): a device selector"""

    def __init__(self, kind, index=0):
        import torch
        self.kind = kind
        self.device = torch.device("cuda", index) if kind == "gpu" else torch.device("cpu")

    def __repr__(self):
        return f"{self.kind}({self.device.index or 0})"


def gpu(index=None):
    """this rank's GPU: LOCAL_RANK under torch.distributed.run (one process per GPU); MST_FORCE_DEVICE pins every rank
    to one card (multi-rank rehearsal on a one-GPU box, with MST_DIST_BACKEND=gloo)"""
    if index is None:
        index = int(os.environ.get("MST_FORCE_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    return Context("gpu", index)


def cpu():
    return Context("cpu")


def host_cpu_share():
    """CPU cores this process may actually use: the cgroup CPU quota when there is one (a GPU box hands each GPU's job a
    16-core share of a 128-core host, and sched_getaffinity still reports all 128), else the affinity mask"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: [t.strip(), None])):
        try:
            quota, period = parse(open(path).read())
            if period is None:
                period = open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()
            if quota not in ("max", "-1"):
                n = min(n, max(1, int(int(quota) / int(period))))
            break
        except (OSError, ValueError):
            continue
    return n


def limit_host_threads():
    """keep torch's intra-op pool within the process's CPU share: oversubscribed pool threads spin when idle and run the
    process into its quota, which stalls the batcher thread for tens of milliseconds at a time"""
    import torch
    n = host_cpu_share()
    if torch.get_num_threads() > n:
        torch.set_num_threads(n)
    return n


class ScalarWriter:
    """what mxboard.SummaryWriter is used for in the reference (trainer.py:84,243-244,257-270): add_scalar(tag, value,
    global_step). mxboard / tensorboard are not dependencies of this build: scalars are appended as JSON lines to
    <logdir>/scalars.jsonl ({"tag", "value", "step", "time"}), which any plotting tool reads."""

    def __init__(self, logdir="/tmp/out"):
        self.path = None
        try:
            os.makedirs(logdir, exist_ok=True)
            self.path = os.path.join(logdir, "scalars.jsonl")
            self._f = open(self.path, "a")
        except OSError:
            self._f = None  # an unwritable log directory must not stop training

    def add_scalar(self, tag, value, global_step=None):
        if self._f is None:
            return
        import json
        import time
        self._f.write(json.dumps({"tag": tag, "value": float(value), "step": global_step, "time": time.time()}) + "\n")

    def flush(self):
        if self._f is not None:
            self._f.flush()

    def close(self):
        if self._f is not None:
            self._f.close()
            self._f = None


def create_directory_if_not_present(path):
    if path and not os.path.exists(path):
        os.makedirs(path, exist_ok=True)


def get_latest_checkpoint_index(model_folder):
    best = None
    for name in os.listdir(model_folder):
        m = re.fullmatch(r"params\.(\d+)(\.npz)?", name)
        if m:
            best = max(best or 0, int(m.group(1)))
    if best is None:
        raise FileNotFoundError("no checkpoint in " + model_folder)
    return best


def save_model(model, path):
    model.save_parameters(path)


def load_model_parameters(model, path, context):
    model.load_parameters(path, context)


def save_object(obj, path):
    with open(path, "wb") as f:
        pickle.dump(obj, f)


def load_object(path):
    with open(path, "rb") as f:
        return pickle.load(f)


def log_config(config):
    import sys
    config.output_to_stream(sys.stdout)


def log_model_variables(model):
    for name, shape in model.store.shapes.items() if model.store is not None else []:
        print(name, shape)
