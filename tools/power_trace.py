"""Is the frame power-bound?  Samples the GPU's hwmon files (package power, power cap, shader clock, temperature) at ~50 Hz while a command runs.

    python tools/power_trace.py [--card N] -- python bench.py --steps 300 --repeats 1 --no-cpu-baseline

Prints one JSON line: the cap, and over the samples taken while the card drew more than half its peak sample (= while the timed regions ran) the mean /
p95 / max power and the mean / min / max shader clock.  Reads sysfs only (an ordinary user may); falls back to `rocm-smi --showpower --showclocks --json`
once per second where the hwmon files are missing.  The sampler runs in THIS process; the command is a child (no GPU call here)."""
import glob
import json
import os
import subprocess
import sys
import time


def hwmons(card):
    """hwmon directories with a power reading: of one card, or of every card (the box shows all GPUs of its host; the one that works is the one that draws)"""
    cands = sorted(glob.glob(f"/sys/class/drm/card{card}/device/hwmon/hwmon*")) if card is not None else sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
    return [h for h in cands if any(os.path.exists(os.path.join(h, f)) for f in ("power1_average", "power1_input"))]


def read_int(path):
    try:
        with open(path) as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return None


def main():
    argv = sys.argv[1:]
    card = None
    if argv and argv[0] == "--card":
        card = int(argv[1]); argv = argv[2:]
    if argv and argv[0] == "--":
        argv = argv[1:]
    hs = hwmons(card)
    h = hs[0] if hs else None
    child = subprocess.Popen(argv)
    samples = []                                           # (t, watts, sclk MHz, temp C) of the card picked below
    per_card = {x: [] for x in hs}
    t0 = time.time()
    smi_every = 1.0
    last_smi = 0.0
    while child.poll() is None:
        now = time.time() - t0
        if h:
            for x in hs:
                p = read_int(os.path.join(x, "power1_average")) or read_int(os.path.join(x, "power1_input"))
                f = read_int(os.path.join(x, "freq1_input"))
                t = read_int(os.path.join(x, "temp1_input")) or read_int(os.path.join(x, "temp2_input"))
                per_card[x].append((now, None if p is None else p / 1e6, None if f is None else f / 1e6, None if t is None else t / 1e3))
            time.sleep(0.02)
        else:
            if now - last_smi >= smi_every:
                last_smi = now
                try:
                    r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10)
                    d = json.loads(r.stdout)
                    c = next(iter(d.values()))
                    pw = next((float(v) for k, v in c.items() if "ower" in k and "(W)" in k), None)
                    sc = next((float(str(v).strip("()Mhz ")) for k, v in c.items() if k.startswith("sclk")), None)
                    samples.append((now, pw, sc, None))
                except Exception:
                    pass
            time.sleep(0.1)
    if per_card:                                           # the card with the highest peak draw is the one the command ran on
        h = max(per_card, key=lambda x: max([s_[1] for s_ in per_card[x] if s_[1] is not None] or [0.0]))
        samples = per_card[h]
    cap = read_int(os.path.join(h, "power1_cap")) if h else None
    watts = [s[1] for s in samples if s[1] is not None]
    out = {"hwmon": h, "cards_sampled": len(per_card), "samples": len(samples), "power_cap_w": None if cap is None else cap / 1e6, "exit_code": child.returncode}
    if watts:
        peak = max(watts)
        busy = [s for s in samples if s[1] is not None and s[1] > 0.5 * peak]
        bw = sorted(s[1] for s in busy)
        bc = [s[2] for s in busy if s[2] is not None]
        out.update({"busy_samples": len(busy), "busy_power_mean_w": round(sum(bw) / len(bw), 1), "busy_power_p95_w": round(bw[int(0.95 * (len(bw) - 1))], 1), "power_max_w": round(peak, 1),
                    "idle_power_w": round(min(watts), 1)})
        if bc:
            out.update({"busy_sclk_mean_mhz": round(sum(bc) / len(bc), 1), "busy_sclk_min_mhz": round(min(bc), 1), "busy_sclk_max_mhz": round(max(bc), 1)})
        temps = [s[3] for s in busy if s[3] is not None]
        if temps:
            out["busy_temp_max_c"] = round(max(temps), 1)
    print("POWER " + json.dumps(out), flush=True)
    return child.returncode


if __name__ == "__main__":
    sys.exit(main())
