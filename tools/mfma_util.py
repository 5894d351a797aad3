"""MFMA pipe utilisation, executed matrix FLOP and the EFFECTIVE CLOCK per kernel from the rocprofv3 pass
`--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE` (tools/profile_round.sh) and, when given, the
tracer's per-kernel average duration (kernel_stats.csv of the `--kernel-trace --stats` pass of the same command):

    tools/mfma_util.py <pmc dir> [kernel_stats.csv]

  cycles per launch = GRBM_GUI_ACTIVE / 8 / launches   (the counter is summed over the 8 XCDs)
  MFMA busy %       = SQ_VALU_MFMA_BUSY_CYCLES / (cycles * 256 CUs * 4 SIMDs)
  executed GFLOP    = SQ_INSTS_VALU_MFMA_MOPS_F16 * 512 / launches       (calibrated on mlp96q_kernel: = its algorithmic FLOP)
  clock             = cycles per launch / tracer duration per launch - what the chip sustained IN that kernel (MI355X_MICROARCH.md, DVFS give-back:
                      1.9 GHz in the attention kernels, not the 2.4 GHz maximum; rounds 1-4 divided by 2 400 MHz here and printed durations 21 % short).
Without a kernel_stats.csv no duration and no clock are printed (a cycle count alone is not a time)."""
import collections
import csv
import glob
import sys


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void w2x::", "").replace("w2x::", "").split("(")[0]


acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[k] += 1
trace_us = {}
if len(sys.argv) > 2:
    for r in csv.DictReader(open(sys.argv[2])):
        try:
            trace_us[short(r["Name"])] = float(r["AverageNs"]) / 1e3
        except (KeyError, ValueError):
            pass
print(f"{'kernel':48s} {'launches':>8s} {'Mcycles':>9s} {'tracer us':>10s} {'clock GHz':>10s} {'MFMA busy %':>12s} {'exec GFLOP':>11s}")
for k, d in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    cyc = d.get("GRBM_GUI_ACTIVE", 0) / 8
    if cyc <= 0 or k.startswith("__amd"):
        continue
    per = cyc / cnt[k]
    us = trace_us.get(k)
    # (the quotient reads high on dispatches shorter than about 0.3 ms: the counter includes the dispatch's ramp; MI355X_MICROARCH.md)
    clock = f"{per / us / 1e3:10.2f}" if us else f"{'-':>10s}"
    print(f"{k[:48]:48s} {cnt[k]:8d} {per / 1e6:9.4f} {us if us else float('nan'):10.1f} {clock} {100 * d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (cyc * 1024):12.1f} "
          f"{d.get('SQ_INSTS_VALU_MFMA_MOPS_F16', 0) * 512 / cnt[k] / 1e9:11.2f}")
