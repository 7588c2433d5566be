import csv, glob, sys
f = sys.argv[1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
prev = None
t0 = int(rows[-n]["Start_Timestamp"])
for r in rows[-n:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {((s - prev) / 1e3) if prev else 0:6.1f}  {r['Kernel_Name'].split('(')[0][-50:]}")
    prev = e
