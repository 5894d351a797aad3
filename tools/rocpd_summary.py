"""Summarise rocprofv3 rocpd (SQLite) output:  rocpd_summary.py <dir-or-db> [kernels|pmc]
kernels: per-kernel calls / total / average / min / max duration (what --stats prints);  pmc: per-kernel counter sums."""
import glob, os, sqlite3, sys

def dbs(path):
    return [path] if path.endswith(".db") else sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))

def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void w2x::", "").replace("w2x::", "").split("(")[0][:70]

def kernels(path):
    rows = {}
    for f in dbs(path):
        cur = sqlite3.connect(f).cursor()
        for name, dur in cur.execute("select name, duration from kernels"):
            r = rows.setdefault(short(name), [0, 0, 1 << 62, 0]); r[0] += 1; r[1] += dur; r[2] = min(r[2], dur); r[3] = max(r[3], dur)
    tot = sum(r[1] for r in rows.values()) or 1
    print("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs")
    for k, r in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        print(f"\"{k}\",{r[0]},{r[1]},{r[1] / r[0]:.0f},{100.0 * r[1] / tot:.2f},{r[2]},{r[3]}")

def pmc(path):
    acc = {}
    for f in dbs(path):
        cur = sqlite3.connect(f).cursor()
        for name, cname, val in cur.execute("select name, counter_name, counter_value from pmc_events"):
            d = acc.setdefault(short(name), {}); e = d.setdefault(cname, [0, 0.0]); e[0] += 1; e[1] += val
    for k, d in sorted(acc.items(), key=lambda kv: -max(v[1] for v in kv[1].values())):
        n = max(v[0] for v in d.values())
        print(k, "dispatches", n)
        for c, (cnt, v) in sorted(d.items()):
            print(f"    {c:28s} {v:18.0f}   per-dispatch {v / cnt:16.0f}")

if __name__ == "__main__":
    (pmc if len(sys.argv) > 2 and sys.argv[2] == "pmc" else kernels)(sys.argv[1])
