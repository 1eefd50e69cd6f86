"""Timeline of the LAST pair of window steps replayed on two streams (tools/gpu_probe_overlap.py under rocprofv3 --kernel-trace):
every kernel of the last ~two Adam-terminated steps with its queue, so that one can see what the second stream's launches did
while the first stream's bag kernels ran.  python tools/gpu_trace_two_streams.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_flat" in r["Kernel_Name"]]
# the concurrent phase is the last thing the probe runs: take the last two Adam kernels (one per stream) and go back to the
# counter bumps that opened their steps
end = adam[-1]
bumps = [i for i in range(end) if "counters_bump" in rows[i]["Kernel_Name"]]
start = bumps[-2]
step = rows[start:end + 1]
t0 = int(step[0]["Start_Timestamp"])
queues = sorted({r["Queue_Id"] for r in step})
print(f"span us {(int(step[-1]['End_Timestamp']) - t0) / 1e3:.1f}, kernels {len(step)}, queues {queues}")
big = ("patch_coattn_fwd", "coattn_bwd8", "patch_wgrad_kernel")
for q in queues:
    mine = [r for r in step if r["Queue_Id"] == q]
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in mine) / 1e3
    print(f"queue {q}: {len(mine)} kernels, {busy:.1f} us of kernel time, first start {(int(mine[0]['Start_Timestamp']) - t0) / 1e3:.1f}, "
          f"last end {(int(mine[-1]['End_Timestamp']) - t0) / 1e3:.1f}")
print("bag kernels and what the other queue ran during each of them:")
for r in step:
    if not any(b in r["Kernel_Name"] for b in big):
        continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    other = [o for o in step if o["Queue_Id"] != r["Queue_Id"] and int(o["Start_Timestamp"]) < e and int(o["End_Timestamp"]) > s]
    ob = sum(min(e, int(o["End_Timestamp"])) - max(s, int(o["Start_Timestamp"])) for o in other) / 1e3
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:34]
    print(f"  q{r['Queue_Id']} {(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:7.1f}  {name:34s} | other queue: {len(other):2d} kernels overlapping, {ob:7.1f} us of them inside this one")
