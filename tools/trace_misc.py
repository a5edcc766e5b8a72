"""Per-step summary of the kernels of a `rocprofv3 --kernel-trace` run of bench.py that are NOT the five GEMM / norm /
attention families: name, launches per step, total and mean duration, mean grid size.
    python tools/trace_misc.py <dir with *_kernel_trace.csv> <steps in the trace>"""
import csv, glob, os, sys, collections
d, steps = sys.argv[1], float(sys.argv[2])
rows = []
for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(p, newline="")))
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in rows:
    n = r["Kernel_Name"]
    if any(k in n for k in ("conv_gemm_kernel", "wgrad_kernel", "gn_", "attn_")):
        continue
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0))
    a = agg[n[:110]]
    a[0] += 1; a[1] += dur; a[2] += grid
tot = 0.0
for n, (c, t, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{t/steps/1e3:7.3f} ms/step {c/steps:7.1f} launches/step  {t/c:8.1f} us avg  grid {g/c:12.0f}  {n}")
    tot += t
print(f"total {tot/steps/1e3:.3f} ms/step")
