import csv, sys, glob
d = sys.argv[1]
f = glob.glob(d + '/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adamw_kernel' in r['Kernel_Name']]
a, b = idx[len(idx)//2], idx[len(idx)//2 + 2]
seg = rows[a+1:b+1]
t0 = int(seg[0]['Start_Timestamp']); t1 = max(int(r['End_Timestamp']) for r in seg)
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg)
# overlap: sweep
ev = []
for r in seg:
    ev.append((int(r['Start_Timestamp']), 1)); ev.append((int(r['End_Timestamp']), -1))
ev.sort()
cur = 0; last = t0; hist = {}
for t, dlt in ev:
    hist[cur] = hist.get(cur, 0) + (t - last); last = t; cur += dlt
print("span us", (t1 - t0) / 1e3, "sum of kernel durations us", busy / 1e3, "kernels", len(seg))
print("time by concurrency:", {k: round(v / 1e3, 1) for k, v in sorted(hist.items())})
qs = {}
for r in seg:
    qs[r.get('Queue_Id', '?')] = qs.get(r.get('Queue_Id', '?'), 0) + 1
print("queues", qs)
