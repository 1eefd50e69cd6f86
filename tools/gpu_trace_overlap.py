"""From a `rocprofv3 --kernel-trace` csv: for every launch of kernel A (substring argv[2]) how many microseconds of it a launch
of kernel B (substring argv[3]) on another queue was running at the same time.  (tools/gpu_probe_dp_overlap.py under rocprofv3.)"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ka, kb = sys.argv[2], sys.argv[3]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), int(r.get("Workgroup_Size_X", 0) or 0),
       int(r.get("Grid_Size_X", 0) or 0)) for r in rows]
ev.sort()
A = [e for e in ev if ka in e[2]]
B = [e for e in ev if kb in e[2]]
print(f"{len(A)} launches of {ka}, {len(B)} of {kb}")
groups = {}
for a in A:
    ov = sum(max(0, min(a[1], b[1]) - max(a[0], b[0])) for b in B if b[3] != a[3])
    wgs = a[5] // max(1, a[4])
    groups.setdefault(wgs, []).append(((a[1] - a[0]) / 1e3, ov / 1e3))
for wgs, v in sorted(groups.items()):
    with_b = [x for x in v if x[1] > 0]
    alone = [x for x in v if x[1] == 0]
    if alone:
        print(f"  {ka} on {wgs} workgroups, nothing beside it: {len(alone)} launches, {sum(x[0] for x in alone) / len(alone):7.1f} us each")
    if with_b:
        print(f"  {ka} on {wgs} workgroups with {kb} beside it: {len(with_b)} launches, {sum(x[0] for x in with_b) / len(with_b):7.1f} us each, "
              f"{sum(x[1] for x in with_b) / len(with_b):7.1f} us of them overlapped")
# and B's durations by whether an A was running
for label, sel in (("alone", lambda b: not any(min(a[1], b[1]) > max(a[0], b[0]) for a in A)), ("beside", lambda b: any(min(a[1], b[1]) > max(a[0], b[0]) for a in A))):
    v = [(b[1] - b[0]) / 1e3 for b in B if sel(b)]
    if v:
        print(f"  {kb} {label} {ka}: {len(v)} launches, {sum(v) / len(v):7.1f} us each")
