"""One step of a rocprofv3 --kernel-trace CSV as a sequence: index, duration (us), gap to the previous kernel's end (us), name.
The step is the span between the last two optimizer launches (k_adam* / k_sgd*)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
idx = [i for i, n in enumerate(names) if 'k_adam' in n or 'k_sgd' in n]
a, b = idx[-2], idx[-1]
tot = busy = 0.0
agg = {}
for i in range(a + 1, b + 1):
    s, e = int(rows[i]['Start_Timestamp']), int(rows[i]['End_Timestamp'])
    d = (e - s) / 1000
    gap = (s - int(rows[i - 1]['End_Timestamp'])) / 1000
    n = names[i].replace('(anonymous namespace)::', '').replace('void ', '')[:90]
    print(f"{i - a:4d} {d:8.1f} gap {gap:6.1f} g{rows[i]['Grid_Size_X']:>8} wg{rows[i]['Workgroup_Size_X']:>4} {n}")
    busy += d
    key = n.split('(')[0]
    c = agg.setdefault(key, [0, 0.0])
    c[0] += 1
    c[1] += d
tot = (int(rows[b]['End_Timestamp']) - int(rows[a]['End_Timestamp'])) / 1000
print(f"# step span {tot:.1f} us, kernel busy {busy:.1f} us, launches {b - a}")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"# {t:9.1f} us {c:4d} x {t / c:8.1f}  {k}")
