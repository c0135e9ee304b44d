"""Turn one round of rocprofv3 output (gpurun_out/<dir>) into the small files committed under profiles/.

usage: make_profile_summary.py <tag> <stats_dir> <fetch_dir> <write_dir> [<bench_json>]
  stats_dir : rocprofv3 --kernel-trace --stats -- python3 bench.py ...
  fetch_dir : rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 bench.py --steps 3 --warmup 1
  write_dir : rocprofv3 --kernel-trace --pmc WRITE_SIZE -- ...      (separate passes, as the HBM section of
              MI355X_MICROARCH.md prescribes; FETCH_SIZE is doubled for gfx950's 64-B tally of 128-B requests; both
              counters are in KiB... reported here in MB = 1e6 bytes after multiplying by 1024)
Writes profiles/<tag>_kernel_stats.csv, <tag>_step_timeline.csv, <tag>_hbm_traffic.csv and merges the per-kernel
traffic into profiles/roofline_traffic.json (read by bench.py for roofline.traffic).
"""
import collections
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, stats_dir, fetch_dir, write_dir = sys.argv[1:5]
out = os.path.join(ROOT, "profiles")


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"_ZN3mst\d+(\w+?)I(DF16b|DF16_|f)?", name)
    if m:
        base = re.match(r"_ZN3mst\d+([a-z_0-9]+?_kernel)", name)
        # integer AND bool template arguments: the bool ones are kernel VARIANTS with their own traffic (attn_bwd_res_kernel<32,0> dense
        # vs <32,1> position-0-sparse; attn_fwd_res_kernel<32,1> with the K|Q|V projection inside)
        tmpl = re.findall(r"L[ib](\d+)E", name)
        if "gemm_nt_kernel" in name or "gemm_nt_pair_kernel" in name:
            tmpl = re.findall(r"Li(\d+)E", name)[:5]  # BM, BN, WGM, WGN, BK (the epilogue switches do not change the traffic key)
        return (base.group(1) if base else m.group(1)) + ("<" + ",".join(tmpl) + ">" if tmpl else "")
    return re.sub(r"\(.*", "", name.replace("mst::", ""))[:70]


shutil.copy(os.path.join(stats_dir, "r_kernel_stats.csv"), os.path.join(out, f"{tag}_kernel_stats.csv"))

# one replayed step: the kernels between two launches of the step's first kernel
rows = sorted(csv.DictReader(open(os.path.join(stats_dir, "r_kernel_trace.csv"))), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "step_begin" in r["Kernel_Name"]]
if len(idx) < 3:  # the piano-roll ends: the bookkeeping rides on the step's first launch, the two embedding GEMMs
    idx = [i for i, r in enumerate(rows) if "gemm_nt_pair_kernel" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
with open(os.path.join(out, f"{tag}_step_timeline.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "grid", "workgroup", "duration_us"])
    tot = 0.0
    for r in rows[a:b]:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tot += d
        w.writerow([short(r["Kernel_Name"]), r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", ""), f"{d:.1f}"])
    w.writerow(["TOTAL (one hipGraph replay incl. the host-side batch copy)", b - a, "", f"{tot:.1f}"])


def counters(d, name):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(os.path.join(d, "r_counter_collection.csv"))):
        if r["Counter_Name"] == name:
            acc[(short(r["Kernel_Name"]), r["Grid_Size"])].append(float(r["Counter_Value"]))
    return {k: sorted(v)[len(v) // 2] for k, v in acc.items()}


fetch, write = counters(fetch_dir, "FETCH_SIZE"), counters(write_dir, "WRITE_SIZE")
traffic = {}
with open(os.path.join(out, f"{tag}_hbm_traffic.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "grid", "fetch_MB(2*FETCH_SIZE)", "write_MB(WRITE_SIZE)", "total_MB_per_launch"])
    for k in sorted(fetch, key=lambda k: -(2 * fetch[k] + write.get(k, 0.0))):
        fm, wm = 2 * fetch[k] * 1024 / 1e6, write.get(k, 0.0) * 1024 / 1e6
        w.writerow([k[0], k[1], f"{fm:.2f}", f"{wm:.2f}", f"{fm + wm:.2f}"])
        traffic[f"{k[0]}@{k[1]}"] = (fm + wm) * 1e6
# keyed by BASELINE config (MST_PROFILE_CONFIG, default 1): bench.py --config N reads configs[N]; only the library's own kernels are kept
cfg = os.environ.get("MST_PROFILE_CONFIG", "1")
path = os.path.join(out, "roofline_traffic.json")
try:
    doc = json.load(open(path))
except (OSError, ValueError):
    doc = {}
if "configs" not in doc:
    doc = {"configs": {}}
own = {k: v for k, v in traffic.items() if not k.startswith(("at::", "__amd", "void at::"))}
doc["configs"][cfg] = {"source": f"profiles/{tag}_hbm_traffic.csv", "bytes_per_launch": own}
json.dump(doc, open(path, "w"), indent=1, sort_keys=True)
if len(sys.argv) > 5:
    shutil.copy(sys.argv[5], os.path.join(out, f"{tag}_bench.json"))
print("wrote", tag, "files;", len(traffic), "kernels with traffic")
