"""For a rocprofv3 --hip-runtime-trace --kernel-trace run: which kernels are launched right after each hipMemcpy* call."""
import csv, glob, sys, collections
d = sys.argv[1]
api = []
for p in glob.glob(d + "/**/*hip_api_trace.csv", recursive=True):
    api += list(csv.DictReader(open(p)))
kern = {}
for p in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        kern[r["Correlation_Id"]] = r["Kernel_Name"][:60]
api.sort(key=lambda r: int(r["Start_Timestamp"]))
ctx = collections.Counter()
names = [r["Function"] for r in api]
for i, r in enumerate(api):
    if r["Function"] in ("hipMemcpyWithStream", "hipMemcpyAsync"):
        nxt = [kern.get(a["Correlation_Id"], "?") for a in api[i + 1:i + 40] if a["Function"] == "hipLaunchKernel"][:1]
        prv = [kern.get(a["Correlation_Id"], "?") for a in api[max(0, i - 40):i] if a["Function"] == "hipLaunchKernel"][-1:]
        ctx[(r["Function"], tuple(prv), tuple(nxt))] += 1
for k, v in ctx.most_common(25):
    print(v, k)
