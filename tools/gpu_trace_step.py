"""Diagnostic: print the timeline of the LAST captured window step from a rocprofv3 --kernel-trace csv."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adam_flat" in r["Kernel_Name"]]
step = rows[idx[-2] + 1: idx[-1] + 1]
# the captured step begins with its counter bump; anything between the previous Adam and that (bench.py's timed roofline
# launch) is not part of it
starts = [i for i, r in enumerate(step) if "counters_bump" in r["Kernel_Name"]]
if starts:
    step = step[starts[-1]:]
t0 = int(step[0]["Start_Timestamp"])
print("step span us", (int(step[-1]["End_Timestamp"]) - t0) / 1e3, "kernels", len(step),
      "sum us", sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step) / 1e3)
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
    print(f'{(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:7.1f} q{r["Queue_Id"]:>2s} grid {r["Grid_Size_X"]:>7s}x{r["Grid_Size_Y"]:>3s}x{r["Grid_Size_Z"]:>2s} {name}')
