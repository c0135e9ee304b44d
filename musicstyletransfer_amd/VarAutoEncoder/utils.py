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


# ---- TensorBoard event files without tensorboard / mxboard: the TFRecord framing (length, masked CRC32C of the length, payload,
# masked CRC32C of the payload) around hand-encoded `Event` protocol-buffer messages. The reference logs through
# mxboard.SummaryWriter(logdir).add_scalar(tag, value, global_step) (trainer.py:84,243-244,257-270); the files written here
# are what that writer produces for scalars, so `tensorboard --logdir` shows the same curves.
_CRC32C_TABLE = []


def _crc32c(data):
    if not _CRC32C_TABLE:
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            _CRC32C_TABLE.append(c)
    crc = 0xFFFFFFFF
    for b in data:
        crc = _CRC32C_TABLE[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def _masked_crc(data):
    c = _crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n):
    out = bytearray()
    n &= (1 << 64) - 1
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _pb_field(number, wire, payload):
    return _varint((number << 3) | wire) + payload


def encode_scalar_event(tag, value, step, wall_time):
    """tensorflow.Event{wall_time = 1 (double), step = 2 (int64), summary = 5 {value = 1 {tag = 1 (string), simple_value = 2
    (float)}}} as bytes"""
    import struct
    t = tag.encode("utf-8")
    val = _pb_field(1, 2, _varint(len(t)) + t) + _pb_field(2, 5, struct.pack("<f", float(value)))
    summary = _pb_field(1, 2, _varint(len(val)) + val)
    return (_pb_field(1, 1, struct.pack("<d", float(wall_time))) + _pb_field(2, 0, _varint(int(step or 0))) +
            _pb_field(5, 2, _varint(len(summary)) + summary))


def tfrecord(payload):
    import struct
    header = struct.pack("<Q", len(payload))
    return header + struct.pack("<I", _masked_crc(header)) + payload + struct.pack("<I", _masked_crc(payload))


class ScalarWriter:
    """what mxboard.SummaryWriter is used for in the reference (trainer.py:84,243-244,257-270): add_scalar(tag, value,
    global_step). Every scalar goes to TWO files under logdir: a TensorBoard event file (events.out.tfevents.<time>.<host>,
    written natively — neither mxboard nor tensorboard is a dependency) and <logdir>/scalars.jsonl ({"tag", "value", "step",
    "time"}), which any plotting tool reads."""

    def __init__(self, logdir="/tmp/out"):
        import socket
        import time
        self.path = self.event_path = None
        self._f = self._ev = None
        try:
            os.makedirs(logdir, exist_ok=True)
            self.path = os.path.join(logdir, "scalars.jsonl")
            self._f = open(self.path, "a")
            self.event_path = os.path.join(logdir, "events.out.tfevents.{:010d}.{}".format(int(time.time()), socket.gethostname()))
            self._ev = open(self.event_path, "ab")
            # the file-version record every event file starts with: Event{wall_time, file_version = 3 (string)}
            import struct
            ver = b"brain.Event:2"
            self._ev.write(tfrecord(_pb_field(1, 1, struct.pack("<d", time.time())) + _pb_field(3, 2, _varint(len(ver)) + ver)))
        except OSError:
            self._f = self._ev = None  # an unwritable log directory must not stop training

    def add_scalar(self, tag, value, global_step=None):
        if self._f is None:
            return
        import json
        import time
        now = time.time()
        self._f.write(json.dumps({"tag": tag, "value": float(value), "step": global_step, "time": now}) + "\n")
        if self._ev is not None:
            self._ev.write(tfrecord(encode_scalar_event(tag, value, global_step, now)))

    def flush(self):
        if self._f is not None:
            self._f.flush()
        if self._ev is not None:
            self._ev.flush()

    def close(self):
        if self._f is not None:
            self._f.close()
            self._f = None
        if self._ev is not None:
            self._ev.close()
            self._ev = None


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
