"""Print the kernel timeline of one all-device frame from a rocprofv3 --kernel-trace csv (arg: trace dir)."""
import csv, glob, sys
if len(sys.argv) < 2:
    raise SystemExit("usage: frame_timeline.py <rocprofv3 output directory with a *kernel_trace.csv>")
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'eskf18_prepare' in r['Kernel_Name'] and i + 1 < len(rows) and 'search_fit' in rows[i + 1]['Kernel_Name']]
i0 = idx[len(idx) // 2]; i1 = idx[len(idx) // 2 + 1]
t0 = int(rows[i0]['Start_Timestamp'])
print("frame period us", (int(rows[i1]['Start_Timestamp']) - t0) / 1e3)
busy = 0
for r in rows[i0 - 2:i1]:
    s = (int(r['Start_Timestamp']) - t0) / 1e3; d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    busy += d
    print(f"{s:8.1f} {d:7.1f} {r['Kernel_Name'][:36]}")
print("gpu busy us", busy)
