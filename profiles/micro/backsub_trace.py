import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
def nm(r):
    return r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0].split('<')[0]
idx = [i for i, r in enumerate(rows) if nm(r) == 'k_build_system']
sel = rows[idx[-1]:]
t0 = int(sel[0]['Start_Timestamp'])
ev = sorted((int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0, nm(r)) for r in sel)
bs = [e for e in ev if e[2].startswith('k_lu_backsub')]
i0 = ev.index(bs[0])
print('span', ev[-1][1]/1e6, 'backsub from', ev[i0-1][0]/1e6)
for e in ev[i0-1:i0+40]: print(f"{e[0]/1e3:10.1f} {e[1]/1e3:10.1f} {(e[1]-e[0])/1e3:7.1f} {e[2]}")
