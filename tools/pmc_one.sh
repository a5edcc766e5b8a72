#!/bin/bash
# PMC of single layers: tools/pmc_one.sh <outdir>   (run through gpurun)
set -o pipefail
OUT=$1; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
for cfg in "wide 1" "narrow 0"; do set -- $cfg
  PSG_WGRAD_WIDE=$2 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq_$1 -o sq -- python3 tools/wgrad_one.py 14 640 640 3 > $OUT/sq_$1.log 2>&1
  PSG_WGRAD_WIDE=$2 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/tcc_$1 -o tcc -- python3 tools/wgrad_one.py 14 640 640 3 > $OUT/tcc_$1.log 2>&1
  PSG_WGRAD_WIDE=$2 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum --output-format csv -d $OUT/tcp_$1 -o tcp -- python3 tools/wgrad_one.py 14 640 640 3 > $OUT/tcp_$1.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/*/")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:40]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    for k, v in agg.items():
        if "wgrad" in k or "conv_gemm" in k:
            print(d.split("/")[-2], k, {a: "%.4g" % b for a, b in v.items()})
PY
