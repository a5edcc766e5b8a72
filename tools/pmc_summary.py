#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes of bench.py into the JSON files bench.py / DESIGN.md cite.

  traffic : python tools/pmc_summary.py traffic  <dir with *_counter_collection.csv of a
            `--pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum` pass>  <workload key>  <out.json>
  sq      : python tools/pmc_summary.py sq <dir of an SQ pass> <out.json>

Memory-side bytes are corrected as MI355X_MICROARCH.md (HBM section) prescribes for gfx950: a 128-byte read request is
tallied as ONE 64-byte request, so bytes = 2 x 64 B x TCC_EA0_RDREQ + 64 B x TCC_EA0_WRREQ (Infinity-Cache hits included).
The build id (sha256[:12] of libpsg_hip.so) ties the summary to the kernels that were profiled; bench.py attaches
`roofline.traffic` only from a summary whose build id and workload match the run."""
import csv
import glob
import hashlib
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def family(name):
    m = re.search(r"conv_gemm_kernelI.*?Li(\d+)ELi(\d+)ELi(\d)E(?:Lb\dE)?EE", name)       # (<T, BM, BN, MODE[, SPLITK]>)
    if m:
        return "conv_gemm(fwd gather)" if m.group(3) in "02" else "conv_gemm(dgrad gather)"
    if "conv_gemm_kernel" in name:
        return "conv_gemm(?)"
    m = re.search(r"conv_pw_kernel<[^>]*?(true|false)>\(|conv_pw_kernelI.*?Lb(\d)EEE", name)      # (<EK, AUX, PRE, TR>: TR names the family)
    if m:
        tr = (m.group(1) == "true") if m.group(1) else (m.group(2) == "1")
        return "conv_gemm(dgrad gather)" if tr else "conv_gemm(fwd gather)"
    if "wgrad_kernel" in name or "wgrad_wide_kernel" in name or "wgrad_pipe_kernel" in name:
        return "wgrad"
    if "attn_" in name:
        return "attention"
    if re.search(r"gn_(stats|apply|bwd|fwd)", name):
        return "groupnorm"
    return None


def short(name):
    m = re.search(r"psg::?(\w+)|_ZN3psg\d+(\w+?)I", name)
    return (m.group(1) or m.group(2)) if m else name[:40]


def rows(d):
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                yield r


def build_id():
    p = os.environ.get("PSG_LIB_PATH") or os.path.join(ROOT, "pokemon_sprite_generator_amd", "libpsg_hip.so")
    return hashlib.sha256(open(p, "rb").read()).hexdigest()[:12]


def traffic(d, workload, out):
    per = defaultdict(lambda: defaultdict(float))      # dispatch -> counter -> value
    fam_of, name_of = {}, {}
    for r in rows(d):
        k = (r["Process_Id"], r["Dispatch_Id"])
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        fam_of[k] = family(r["Kernel_Name"])
        name_of[k] = short(r["Kernel_Name"])
    fams = defaultdict(lambda: [0, 0.0, 0.0])
    kern = defaultdict(lambda: [0, 0.0, 0.0])
    total = 0.0
    for k, c in per.items():
        rd = 2 * 64.0 * c.get("TCC_EA0_RDREQ_sum", 0.0)
        wr = 64.0 * c.get("TCC_EA0_WRREQ_sum", 0.0)
        total += rd + wr
        for table, key in ((fams, fam_of[k]), (kern, name_of[k])):
            if key:
                table[key][0] += 1; table[key][1] += rd; table[key][2] += wr
    js = {"source": "rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -- python3 bench.py ... (tools/pmc_summary.py)",
          "correction": "bytes = 2 x 64 B x TCC_EA0_RDREQ (gfx950: 128-byte read requests are tallied at 64 B) + 64 B x TCC_EA0_WRREQ; "
                        "Infinity-Cache hits are included (MI355X_MICROARCH.md, HBM section)",
          "build_id": build_id(), "workload": workload,
          "families": {f: {"launches_profiled": n, "read_bytes_per_launch": rd / n, "write_bytes_per_launch": wr / n,
                           "traffic_bytes_per_launch": (rd + wr) / n} for f, (n, rd, wr) in fams.items()},
          "kernels": {f: {"launches_profiled": n, "traffic_bytes_per_launch": (rd + wr) / n, "total_GB": (rd + wr) / 1e9}
                      for f, (n, rd, wr) in sorted(kern.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:40]},
          "all_dispatches_bytes": total}
    json.dump(js, open(out, "w"), indent=1)
    print(json.dumps({f: round(v["traffic_bytes_per_launch"] / 1e6, 1) for f, v in js["families"].items()}), "MB/launch; total", round(total / 1e9, 1), "GB")


def sq(d, out):
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(set)
    for r in rows(d):
        n = short(r["Kernel_Name"])
        agg[n][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[n].add((r["Process_Id"], r["Dispatch_Id"]))
    res = {}
    for n, c in agg.items():
        e = {"launches": len(cnt[n]), **{k: v for k, v in c.items()}}
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        if wc:
            # SQ_WAVE_CYCLES / WAIT / ACTIVE count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES counts cycles (MI355X_MICROARCH.md)
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
                if k in c:
                    e[k + "_frac_of_wave_cycles"] = c[k] / wc
        if c.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_bank_conflict_frac"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
        if c.get("TCC_HIT_sum") is not None and (c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)) > 0:
            e["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
        res[n] = e
    json.dump({"build_id": build_id(), "kernels": res}, open(out, "w"), indent=1)
    for n, e in sorted(res.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:12]:
        print(n, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in e.items() if "frac" in k or k in ("launches", "l2_hit_rate")})


if __name__ == "__main__":
    if sys.argv[1] == "traffic":
        traffic(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        sq(sys.argv[2], sys.argv[3])
