"""Per hardware queue: kernels and busy time of the LAST frame in a rocprofv3 kernel-trace csv (tools/ab/render_trace.py)."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t_end = int(rows[-1]["End_Timestamp"])
last = [r for r in rows if int(r["Start_Timestamp"]) > t_end - 9_000_000]      # the last 9 ms
q = collections.defaultdict(lambda: [0, 0, collections.Counter()])
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    e = q[r["Queue_Id"]]; e[0] += 1; e[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); e[2][r["Kernel_Name"].split("(")[0][-40:]] += 1
for k, (n, ns, names) in sorted(q.items()):
    print(f"queue {k}: {n} kernels, {ns / 1e6:.2f} ms busy; {dict(names.most_common(4))}")
for r in last:
    if "rocclr" in r["Kernel_Name"] or "compose" in r["Kernel_Name"]:
        print(f"  {(int(r['Start_Timestamp']) - t0) / 1e6:7.3f} .. {(int(r['End_Timestamp']) - t0) / 1e6:7.3f} ms  queue {r['Queue_Id']}  {r['Kernel_Name'][:60]}")
