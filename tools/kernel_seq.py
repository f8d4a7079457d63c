"""Dev: print the last N kernels of a rocprofv3 --kernel-trace csv in start order with their durations (which launch sits next to which)."""
import csv, glob, sys
root, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 300
f = glob.glob(f"{root}/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
for r in rows[-n:]:
    print(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us  {r['Kernel_Name'][:100]}")
