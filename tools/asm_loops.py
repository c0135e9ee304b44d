"""Per-loop instruction mix of the gfx950 code of one csrc/*.hip file (not a test; runs without a GPU).

    python tools/asm_loops.py attention.hip [mangled-name-substring ...]

Compiles the file to assembly with the flags musicstyletransfer_amd/csrc/build.py uses for it and, for every kernel whose mangled
name contains one of the substrings (all kernels without any), lists each loop that holds an MFMA — the instructions between a label
and a backward branch to it, so an outer loop includes its inner ones: MFMAs, VALU issue slots (a transcendental counts twice: 8 cycles
against 4), moves / permutes, conversions, integer divisions (v_cvt_f32_u32 sequences), global / LDS / scalar loads, branches, scratch
traffic — plus the kernel's register and spill counts. This is how the notes' "issue slots per tile" figures were taken."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from musicstyletransfer_amd.csrc import build  # noqa: E402


def assembly(src):
    out = os.path.join(tempfile.mkdtemp(prefix="mst_asm_"), src.replace(".hip", ".s"))
    cmd = [build.hipcc(), *[f for f in build.FLAGS if f != "-fPIC"], *build.FILE_FLAGS.get(src, []), "-S", "--cuda-device-only",
           os.path.join(build.HERE, src), "-o", out]
    subprocess.run(cmd, check=True, capture_output=True)
    return open(out).read()


def main():
    src, want = sys.argv[1], sys.argv[2:]
    text = assembly(src)
    regs = {m.group(1): (m.group(2), m.group(3)) for m in
            re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", text)}
    lines = text.split("\n")
    starts = [(i, ln.split(":")[0]) for i, ln in enumerate(lines) if re.match(r"^_Z\w+:", ln)] + [(len(lines), "")]
    for (s, name), (e, _) in zip(starts, starts[1:]):
        if want and not any(w in name for w in want):
            continue
        body = lines[s:e]
        labels = {m.group(1): i for i, ln in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", ln))}
        print(f"== {name}  vgpr {regs.get(name, ('?', '?'))[0]} spilled {regs.get(name, ('?', '?'))[1]}")
        for i, ln in enumerate(body):
            m = re.match(r"\s+s_c?branch\w* (\.LBB\d+_\d+)", ln)
            if not (m and m.group(1) in labels and labels[m.group(1)] < i):
                continue
            seg = body[labels[m.group(1)]:i]
            c = collections.Counter(x.split()[0] for x in seg if re.match(r"^\s+[vsdgb]\w+", x))
            tot = lambda pred: sum(v for k, v in c.items() if pred(k))  # noqa: E731
            mfma = tot(lambda k: k.startswith("v_mfma"))
            if not mfma:
                continue
            trans = tot(lambda k: re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_", k) is not None)
            valu = tot(lambda k: k.startswith("v_") and not k.startswith("v_mfma")) + trans
            print(f"  loop@{labels[m.group(1)]:6d} lines {len(seg):5d} mfma {mfma:3d} valu-slots {valu:5d} (transc {trans:3d}, mov/perm "
                  f"{c['v_mov_b32_e32'] + c['v_perm_b32'] + c['v_alignbit_b32']:3d}, cvt {tot(lambda k: k.startswith('v_cvt_pk')):3d}, "
                  f"div {c['v_cvt_f32_u32_e32']:2d}) salu {tot(lambda k: k.startswith('s_') and k not in ('s_waitcnt', 's_nop')):4d} "
                  f"gload {tot(lambda k: k.startswith(('global_load', 'buffer_load'))):3d} lds-read {tot(lambda k: k.startswith('ds_read')):3d} "
                  f"s_load {tot(lambda k: k.startswith('s_load')):3d} branches {tot(lambda k: k.startswith('s_cbranch')):3d} "
                  f"scratch {tot(lambda k: k.startswith('scratch')):3d}")


if __name__ == "__main__":
    main()
