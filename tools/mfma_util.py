"""MFMA pipe utilisation per kernel from the rocprofv3 pass `--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES ... GRBM_GUI_ACTIVE`
(tools/profile_round.sh):  util = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles * 256 CUs * 4 SIMDs), kernel cycles =
GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs; cross-checked against the kernel-trace durations at 2.4 GHz)."""
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void w2x::", "").replace("w2x::", "").split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE": cnt[k] += 1
print(f"{'kernel':48s} {'launches':>8s} {'us/launch':>10s} {'MFMA busy %':>12s}")
for k, d in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    cyc = d.get("GRBM_GUI_ACTIVE", 0) / 8
    if cyc <= 0 or k.startswith("__amd"): continue
    print(f"{k[:48]:48s} {cnt[k]:8d} {cyc / cnt[k] / 2400:10.1f} {100 * d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (cyc * 1024):12.1f}")
