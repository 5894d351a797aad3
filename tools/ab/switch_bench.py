"""bench.py with library debug switches set (w2x_debug_set: the reference paths have no environment names - INTEGRATION.md section 4), alternating:

    python tools/ab/switch_bench.py ROUNDS "" "no_fuse_stem=1" ... [-- bench.py arguments, e.g. --config 2]

One line per run: the setting, the median resident frame time and its samples, and the per-kernel milliseconds of a frame.  Each run is a child process
(the switches are process-wide and read at load())."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = """
import importlib, runpy, sys
sys.path.insert(0, {root!r})
pkg = importlib.import_module("waifu2x-tensorrt_amd")
for kv in {setting!r}.split():
    k, v = kv.split("=")
    assert pkg.lib().w2x_debug_set(k.encode(), int(v)), k
sys.argv = ["bench.py"] + {args!r}
runpy.run_path({bench!r}, run_name="__main__")
"""


def main():
    argv = sys.argv[1:]
    extra = []
    if "--" in argv:
        k = argv.index("--")
        argv, extra = argv[:k], argv[k + 1:]
    rounds, settings = int(argv[0]), argv[1:]
    args = ["--steps", os.environ.get("STEPS", "20"), "--warmup", "3", "--no-cpu-baseline"] + extra
    for _ in range(rounds):
        for s in settings:
            code = CHILD.format(root=ROOT, setting=s, args=args, bench=os.path.join(ROOT, "bench.py"))
            r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT)
            try:
                d = json.loads(r.stdout.strip().splitlines()[-1])
                print(f"[{s}] ms/frame {d['ms_per_step']} {d.get('ms_per_step_samples')} | {d['roofline'].get('kernels_ms_per_frame')}", flush=True)
            except Exception:
                print(f"[{s}] failed: {r.stderr[-400:]}", flush=True)


if __name__ == "__main__":
    main()
