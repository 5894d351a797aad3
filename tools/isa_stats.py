"""Static ISA statistics of one kernel source:  python tools/isa_stats.py waifu2x-tensorrt_amd/csrc/k_swinattn96.hip [--flags "..."] [--top 30]

Compiles the file for gfx950 to assembly (device only) and prints, per kernel: VGPR / SGPR counts, spills, scratch bytes,
LDS bytes, and a histogram of the instruction mnemonics (VALU vs MFMA vs LDS vs global).  This is the view the "VALU diet"
of the attention kernels was done with: the kernels are VALU-issue bound, so the instruction mix
around the MFMAs is what gets optimised.  Static counts: branches not taken at run time (e.g. the LayerNorm statistics
epilogue when no consumer needs them) are included."""
import argparse, collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("source")
    ap.add_argument("--flags", default="")
    ap.add_argument("--top", type=int, default=24)
    ap.add_argument("--kernel", default="", help="substring of the (mangled) kernel name")
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = ["hipcc", "-std=c++17", "-O3", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-I", os.path.join(ROOT, "include"),
               "-I", os.path.join(ROOT, "waifu2x-tensorrt_amd", "csrc"), *a.flags.split(), "-o", out, a.source]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stderr)
        text = open(out).read()
    # kernel bodies: from "<name>:" to s_endpgm
    bodies = {}
    cur = None
    for line in text.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1); bodies[cur] = []
        elif cur is not None:
            bodies[cur].append(line)
            if "s_endpgm" in line:
                cur = None
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S+)(.*?)\.wavefront_size", text, re.S):
        d = dict(re.findall(r"\.(\w+):\s+(\d+)", m.group(2)))
        meta[m.group(1)] = d
    for name, body in bodies.items():
        if a.kernel and a.kernel not in name:
            continue
        d = meta.get(name, {})
        hist = collections.Counter()
        for line in body:
            m = re.match(r"^\s+(v_\w+|s_\w+|ds_\w+|global_\w+|buffer_\w+|scratch_\w+|flat_\w+)", line)
            if m:
                hist[m.group(1)] += 1
        valu = sum(n for k, n in hist.items() if k.startswith("v_") and not k.startswith("v_mfma"))
        mfma = sum(n for k, n in hist.items() if k.startswith("v_mfma"))
        print(f"{name}\n  vgpr {d.get('vgpr_count')}  sgpr {d.get('sgpr_count')}  spilled vgprs {d.get('vgpr_spill_count')}  scratch {d.get('private_segment_fixed_size')} B"
              f"  static LDS {d.get('group_segment_fixed_size')} B\n  VALU {valu}  MFMA {mfma}  (VALU per MFMA {valu / max(mfma, 1):.1f})  s_nop {hist.get('s_nop', 0)}"
              f"  LDS ops {sum(n for k, n in hist.items() if k.startswith('ds_'))}  global ops {sum(n for k, n in hist.items() if k.startswith('global_'))}")
        for k, n in hist.most_common(a.top):
            print(f"    {n:5d}  {k}")


if __name__ == "__main__":
    main()
