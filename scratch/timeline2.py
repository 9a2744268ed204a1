import csv, sys, glob
d = sys.argv[1]
f = glob.glob(d + '/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adamw_kernel' in r['Kernel_Name']]
a, b = idx[len(idx)//2], idx[len(idx)//2 + 1]
t0 = int(rows[a + 1]['Start_Timestamp'])
prev_end = None
for r in rows[a + 1:b + 2]:
    s = int(r['Start_Timestamp']) - t0; e = int(r['End_Timestamp']) - t0
    gap = s - prev_end if prev_end is not None else 0
    prev_end = e
    nm = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    print(f"{s/1e3:8.1f} dur={(e-s)/1e3:6.1f} gap={gap/1e3:5.1f} {nm[:90]}")
