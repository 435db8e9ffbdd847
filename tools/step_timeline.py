"""One training step as a kernel timeline, from a `rocprofv3 --kernel-trace --output-format csv` run of tools/step_prof.py
(tools/prof_step.sh leaves it in gpurun_out/prof_<tag>/p_kernel_trace.csv).  The LAST step between two adam_kernel launches: per
launch its start (ms since the step's first kernel), duration, the idle gap in front of it (against the latest end seen so far: what
a one-stream run leaves between kernels), grid and the kernel's name; then the sums of kernel time and of gaps.
usage: python tools/step_timeline.py gpurun_out/prof_<tag>/p_kernel_trace.csv [min_us=0] [first=<kernel-name part>] > timeline.txt
`first=...`: units begin WITH a launch of that kernel instead of ending with adam_kernel (e.g. first=prepare_image_kernel: one CRF call)."""
import csv
import re
import sys

path = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
first = next((a.split("=", 1)[1] for a in sys.argv[3:] if a.startswith("first=")), None)
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if (first or "adam_kernel") in r["Kernel_Name"]]
if len(marks) < 2:
    sys.exit("fewer than two units in the trace")
step = rows[marks[-2]:marks[-1]] if first else rows[marks[-2] + 1:marks[-1] + 1]
t0 = int(step[0]["Start_Timestamp"])
last_end = t0
busy = gaps = 0
short = lambda n: re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void ", "", n))[:110]
print(f"# {len(step)} launches, wall {(int(step[-1]['End_Timestamp']) - t0) / 1e6:.3f} ms")
print("# start_ms   dur_us   gap_us  grid(workgroups)  kernel")
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = max(0, s - last_end)
    wg = 1
    for ax in "XYZ":
        g, w = int(r.get(f"Grid_Size_{ax}", 1) or 1), int(r.get(f"Workgroup_Size_{ax}", 1) or 1)
        wg *= max(1, g // max(1, w))
    busy += e - s
    gaps += gap
    if (e - s) / 1e3 >= min_us or gap / 1e3 >= 5:
        print(f"{(s - t0) / 1e6:9.3f} {(e - s) / 1e3:8.1f} {gap / 1e3:8.1f} {wg:9d}  {short(r['Kernel_Name'])}")
    last_end = max(last_end, e)
print(f"# kernel time {busy / 1e6:.3f} ms, idle in front of kernels {gaps / 1e6:.3f} ms")
