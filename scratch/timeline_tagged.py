"""One replay of the tagged step out of a rocprofv3 kernel trace: per-queue busy time, concurrency, and the launches in order."""
import csv, sys, glob, collections
d = sys.argv[1]
f = glob.glob(d + '/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adamw_kernel' in r['Kernel_Name']]
a, b = idx[len(idx) // 2], idx[len(idx) // 2 + 1]
sel = rows[a + 1:b + 1]
t0 = int(sel[0]['Start_Timestamp'])
span = int(sel[-1]['End_Timestamp']) - t0
busy = collections.defaultdict(int); cnt = collections.Counter()
qkey = 'Queue_Id' if 'Queue_Id' in sel[0] else 'Stream_Id'
ev = []
for r in sel:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    busy[r[qkey]] += e - s; cnt[r[qkey]] += 1
    ev += [(s, 1), (e, -1)]
ev.sort()
lvl, last, hist = 0, 0, collections.defaultdict(int)
for t, dl in ev:
    hist[lvl] += t - last; last = t; lvl += dl
print(f"launches {len(sel)}  span {span/1e3:.1f} us  sum of durations {sum(busy.values())/1e3:.1f} us")
for q in busy: print(f"  queue {q}: {cnt[q]} launches, busy {busy[q]/1e3:.1f} us")
print("  time with k kernels in flight:", {k: round(v / 1e3, 1) for k, v in sorted(hist.items())})
if len(sys.argv) > 2:
    for r in sel:
        s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
        nm = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
        print(f"{s/1e3:8.1f} dur={(e-s)/1e3:6.1f} q={r[qkey]} grid={r['Grid_Size_X']:>7s}x{r['Grid_Size_Y']} wg={r['Workgroup_Size_X']:>4s} {nm[:70]}")
