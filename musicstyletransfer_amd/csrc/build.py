"""Build libmst_hip.so (gfx950) in-tree with hipcc. No torch headers are involved: the library is a
plain C-ABI shared object (include/mst_hip.h) that the Python host loads with ctypes.

    python -m musicstyletransfer_amd.csrc.build [--force] [--jobs N]
"""
import argparse
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libmst_hip.so")
OBJ_DIR = os.path.join(HERE, "_obj")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-Wno-unused-but-set-variable"]
# Per-file code generation switches. attention.hip: MFMA results feed VALU work (exp, scaling) at once, so the
# accumulators should live in VGPRs, not AGPRs: no v_accvgpr_read copies and 150 instead of 184 registers per
# lane (3 waves/SIMD instead of 2).
# -fno-slp-vectorize: left to itself hipcc pairs the per-element fp32 multiplies of a probability tile into v_pk_mul_f32
# one register off the MFMA accumulator's pairs (fourteen v_mov + eight v_alignbit / v_perm per tile to repair them), and
# packed fp32 beside MFMAs costs more issue time than the two scalar instructions it replaces anyway
# (MI355X_MICROARCH: "an anti-lever beside MFMAs"). In-call A/B, scalar vs hand-packed vs compiler-packed: kernel_notes.
FILE_FLAGS = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-slp-vectorize"]}
for _kv in os.environ.get("MST_EXTRA_FLAGS", "").split(";"):  # "file.hip=-flag -flag;..." for experiments
    if "=" in _kv:
        FILE_FLAGS.setdefault(_kv.split("=", 1)[0], []).extend(_kv.split("=", 1)[1].split())


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def sources():
    return sorted(f for f in os.listdir(HERE) if f.endswith(".hip"))


def headers():
    hs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".hpp")]
    hs.append(os.path.join(HERE, "..", "..", "include", "mst_hip.h"))
    hs.append(os.path.abspath(__file__))  # flags live here
    return hs


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src):
    obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
    cmd = [hipcc(), *FLAGS, *FILE_FLAGS.get(src, []), "-c", os.path.join(HERE, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force=False, jobs=4, verbose=True):
    os.makedirs(OBJ_DIR, exist_ok=True)
    hdrs = headers()
    todo, objs = [], []
    for s in sources():
        obj = os.path.join(OBJ_DIR, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [os.path.join(HERE, s), *hdrs]):
            todo.append(s)
    if todo:
        if verbose:
            print(f"[mst build] hipcc --offload-arch={ARCH}: {', '.join(todo)}", flush=True)
        with ThreadPoolExecutor(max_workers=max(1, jobs)) as ex:
            list(ex.map(_compile, todo))
    if force or todo or _stale(LIB, objs):
        cmd = [hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[mst build] linked {LIB}", flush=True)
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=4)
    a = ap.parse_args()
    build(force=a.force, jobs=a.jobs)
