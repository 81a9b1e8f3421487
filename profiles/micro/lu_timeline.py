"""Summarises a rocprofv3 kernel trace of profiles/micro/fit_only.py: span of the last solve, leaf / panel-update totals, the gaps
between panels (critical path outside the leaf chain) and the back substitution span."""
import csv, sys, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
def nm(r):
    return r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0].split('<')[0]
idx = [i for i, r in enumerate(rows) if nm(r) == 'k_build_system']
sel = rows[idx[-1]:]
t0 = int(sel[0]['Start_Timestamp'])
ev = sorted((int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0, nm(r)) for r in sel)
print('solve span ms', max(e[1] for e in ev) / 1e6)
tot = {}
for e in ev: tot[e[2]] = tot.get(e[2], 0) + (e[1] - e[0]) / 1e6
print({k: round(v, 2) for k, v in sorted(tot.items(), key=lambda x: -x[1])})
leaf = [e for e in ev if e[2] == 'k_lu_leaf']
gaps = [(b[0] - a[1]) / 1e3 for a, b in zip(leaf[:-1], leaf[1:]) if (b[0] - a[1]) / 1e3 > 25]
print('inter-panel gaps', len(gaps), 'median us', statistics.median(gaps), 'sum ms', sum(gaps) / 1e3)
upd = [i for i, e in enumerate(ev) if e[2].startswith('k_lu_backsub')]
first_bs = max(i for i in range(upd[0]) if ev[i][2].startswith('k_lu_trsm'))       # the triangular solve in front of the first update
print('back substitution span ms', (ev[-1][1] - ev[first_bs][0]) / 1e6, ' first leaf at ms', leaf[0][0] / 1e6, ' last leaf end ms', leaf[-1][1] / 1e6)
