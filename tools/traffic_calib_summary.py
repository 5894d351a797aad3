"""Calibration table of the HBM counters: counter value (KiB -> bytes) over the known byte count of each micro-kernel of
tools/traffic_calib.hip.   usage: traffic_calib_summary.py <stdout of traffic_calib> <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass>"""
import collections, csv, glob, re, sys

known = {}
for line in open(sys.argv[1]):
    m = re.match(r"KNOWN (\S+) read (\d+) write (\d+)", line)
    if m: known[m.group(1)] = (int(m.group(2)), int(m.group(3)))

def counters(d, name):
    acc = collections.defaultdict(float)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name: acc[r["Kernel_Name"]] += float(r["Counter_Value"])
    return acc

fetch, write = counters(sys.argv[2], "FETCH_SIZE"), counters(sys.argv[3], "WRITE_SIZE")
print(f"{'micro-kernel (access shape)':44s} {'known read MB':>14s} {'FETCH_SIZE MB':>14s} {'ratio':>7s} {'known write MB':>15s} {'WRITE_SIZE MB':>14s} {'ratio':>7s}")
for k, (rd, wr) in known.items():
    base = k.split("<")[0]
    def find(acc):
        for name, v in acc.items():
            if base in name and (("<" not in k) or k.split("<")[1].rstrip(">").replace(",", ", ") in name or k.split("<")[1].rstrip(">") in name.replace(" ", "")): return v * 1024
        return float("nan")
    f, w = find(fetch), find(write)
    print(f"{k:44s} {rd / 1e6:14.1f} {f / 1e6:14.1f} {(f / rd if rd else float('nan')):7.3f} {wr / 1e6:15.1f} {w / 1e6:14.1f} {(w / wr if wr else float('nan')):7.3f}")
