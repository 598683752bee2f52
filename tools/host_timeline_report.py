"""Reads rocprofv3's kernel_trace.csv and memory_copy_trace.csv of tools/host_decode_timeline.py and prints the device-side timeline of the LAST
host-pointer decode call: every copy and kernel with start / end relative to the call's first copy, and the idle gaps of the D2H direction."""
import csv
import glob
import sys

d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", r.get("Name", "copy")), ""))
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "kernel", r["Kernel_Name"][:60]))
ev.sort()
# the last call: events after the last gap of more than 30 ms
cut = 0
for i in range(1, len(ev)):
    if ev[i][0] - max(e[1] for e in ev[:i][-8:]) > 30_000_000: cut = i
call = ev[cut:]
t0 = call[0][0]
last_d2h_end = None
for s, e, kind, name in call:
    gap = ""
    if "DEVICE_TO_HOST" in kind.upper() or "D2H" in kind.upper():
        if last_d2h_end is not None: gap = "   (D2H idle before: %.2f ms)" % ((s - last_d2h_end) / 1e6)
        last_d2h_end = e
    print("%8.2f .. %8.2f ms  %6.2f ms  %-28s %s%s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, kind, name, gap))
print("device-side span of the call: %.2f ms" % ((max(e for _, e, _, _ in call) - t0) / 1e6))
