"""kernels of the last proof in a rocprofv3 kernel-trace CSV: start, duration, gap to the previous kernel (us)"""
import csv, sys, glob
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
last = rows[-n:]
t0 = int(last[0]["Start_Timestamp"]); prev = None
for r in last:
    s = int(r["Start_Timestamp"]); e = int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void arkbp::", "")[:36]
    print("%9.1f  dur %7.1f  gap %7.1f  %-36s grid %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0, name, r.get("Grid_Size_X", "")))
    prev = e
