"""Per-kernel totals and the timeline of the LAST ridge solve in a rocprofv3 --kernel-trace CSV (profiles/micro/fit_solvers.py chol N):
python trace_summary.py <kernel_trace.csv> [first_n_rows]"""
import csv, sys, collections
tr = list(csv.DictReader(open(sys.argv[1])))
nshow = int(sys.argv[2]) if len(sys.argv) > 2 else 0
idx = [i for i, r in enumerate(tr) if 'k_build_system' in r['Kernel_Name']]
seg = tr[idx[-1]:]
# the solve ends with its last own kernel (what follows in the trace is the script's residual check)
own = ('k_extract_wout', 'k_symmetrize', 'k_lu_', 'k_chol', 'k_gemm_nt_dma', 'k_gemm_acc', 'k_build_system')
last = max(i for i, r in enumerate(seg) if any(k in r['Kernel_Name'] for k in own))
seg = seg[:last + 1]
t0 = int(seg[0]['Start_Timestamp'])
end = max(int(r['End_Timestamp']) for r in seg)
print(f"last solve: {len(seg)} launches, span {(end - t0) / 1e6:.3f} ms")
tot = collections.defaultdict(lambda: [0, 0.0])
for r in seg:
    name = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:40]
    tot[name][0] += 1
    tot[name][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for k, (n, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:34s} {n:5d} launches {us / 1e3:8.3f} ms  avg {us / n:7.1f} us")
for r in seg[:nshow]:
    print(f"  {r['Kernel_Name'][:48]:48s} stream {r['Stream_Id']:>3s} start {(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} us  grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}")
