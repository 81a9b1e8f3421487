"""Kernel by kernel through one inter-panel gap of the single ridge solve (mid-matrix), from a rocprofv3 kernel trace of fit_only.py."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
def nm(r):
    return r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0].split('<')[0]
idx = [i for i, r in enumerate(rows) if nm(r) == 'k_build_system']
sel = rows[idx[-1]:]
t0 = int(sel[0]['Start_Timestamp'])
ev = sorted((int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0, nm(r), r.get('Queue_Id', '?')) for r in sel)
perm = [i for i, e in enumerate(ev) if e[2] == 'k_lu_perm_src']
i0 = perm[len(perm) // 2]
for e in ev[i0 - 4:i0 + 16]: print(f"{e[0]/1e3:10.1f} {e[1]/1e3:10.1f} {(e[1]-e[0])/1e3:7.1f} q{e[3]} {e[2]}")
