#!/bin/bash
# PMC of single weight-gradient layers: tools/pmc_one.sh <outdir> [H Cin Cout ks]   (run through gpurun)
set -o pipefail
OUT=$1; mkdir -p $OUT; export TMPDIR=/tmp
L="${2:-14} ${3:-640} ${4:-640} ${5:-3}"
for cfg in "pipe 1 1" "wide 0 1" "narrow 0 0"; do set -- $cfg
  PSG_WGRAD_PIPE=$2 PSG_WGRAD_WIDE=$3 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq_$1 -o sq -- python3 tools/wgrad_one.py $L > $OUT/sq_$1.log 2>&1
  PSG_WGRAD_PIPE=$2 PSG_WGRAD_WIDE=$3 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum TCC_HIT_sum --output-format csv -d $OUT/tcp_$1 -o tcp -- python3 tools/wgrad_one.py $L > $OUT/tcp_$1.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/*/")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:40]][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in agg.items():
        if "wgrad" in k and "sum" not in k:
            print(d.split("/")[-2], k, {a: "%.4g" % b for a, b in v.items()})
PY
