"""Per-stream busy time of the LAST batched ridge solve in a rocprofv3 --kernel-trace CSV (profiles/micro/fit_batch16.py):
python stream_busy.py <kernel_trace.csv>   -- which stream is the long one, and what runs on it."""
import csv, sys, collections
tr = list(csv.DictReader(open(sys.argv[1])))
idx = [i for i, r in enumerate(tr) if 'k_build_system' in r['Kernel_Name']]
seg = tr[idx[-1]:]
last = max(i for i, r in enumerate(seg) if 'k_extract_wout' in r['Kernel_Name'] or 'k_symmetrize' in r['Kernel_Name'])
seg = seg[:last + 1]
t0 = min(int(r['Start_Timestamp']) for r in seg); t1 = max(int(r['End_Timestamp']) for r in seg)
print(f"solve: {len(seg)} launches, span {(t1 - t0) / 1e6:.3f} ms")
per = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in seg:
    name = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:28]
    e = per[r['Stream_Id']][name]
    e[0] += 1; e[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for sid, d in per.items():
    print(f"stream {sid}: busy {sum(v[1] for v in d.values()) / 1e3:.3f} ms")
    for k, (n, us) in sorted(d.items(), key=lambda kv: -kv[1][1]):
        print(f"    {k:30s} {n:5d} launches {us / 1e3:8.3f} ms  avg {us / n:7.1f} us")
