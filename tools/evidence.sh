#!/bin/bash
# Evidence pass for one build (run on the GPU box through gpurun): bench lines, rocprofv3 kernel stats, the two PMC passes.
#   tools/evidence.sh r02_a            -> gpurun_out/ev_r02_a/*  (copy what should be judged into profiles/)
# Every GPU step is joined with && (a step that fails or times out ends the pass); rocprofv3 gets python3 itself after --.
set -e -o pipefail
TAG=${1:-r02_x}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/ev_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/${TAG}_bench_bs256_bf16.json 2> $OUT/bench.err
echo "bench done"; cat $OUT/${TAG}_bench_bs256_bf16.json
export PSG_WGRAD_STREAM=0     # (the default since round 3; explicit so that per-kernel times are exclusive)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o $TAG -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile --no-secondary > $OUT/prof.log 2>&1
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_bs256_bf16_kernel_stats.csv
echo "kernel stats done"
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $OUT/pmc_tcc -o tcc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-secondary > $OUT/pmc_tcc.log 2>&1
python tools/pmc_summary.py traffic $OUT/pmc_tcc train_bf16_bs256 $OUT/${TAG}_pmc_traffic.json > $OUT/pmc_tcc_summary.txt
echo "tcc done"
# the same counters for the two secondary workloads of the default line (their `roofline.traffic`)
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $OUT/pmc_tcc_s -o tccs -- python3 bench.py --mode sample --no-graph --steps 3 --warmup 2 --no-cpu-baseline --no-profile > $OUT/pmc_tcc_s.log 2>&1
python tools/pmc_summary.py traffic $OUT/pmc_tcc_s sample_bf16_n64 $OUT/${TAG}_pmc_traffic_sample.json > $OUT/pmc_tcc_s_summary.txt
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $OUT/pmc_tcc_f -o tccf -- python3 bench.py --config fwd_bwd_fp32_bs64 --steps 2 --warmup 1 --no-cpu-baseline --no-profile > $OUT/pmc_tcc_f.log 2>&1
python tools/pmc_summary.py traffic $OUT/pmc_tcc_f fwd_bwd_fp32_bs64 $OUT/${TAG}_pmc_traffic_fp32.json > $OUT/pmc_tcc_f_summary.txt
echo "tcc secondary done"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq -o sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-secondary > $OUT/pmc_sq.log 2>&1
python tools/pmc_summary.py sq $OUT/pmc_sq $OUT/${TAG}_pmc_sq.json > $OUT/pmc_sq_summary.txt
echo "sq done"
unset PSG_WGRAD_STREAM
python bench.py --mode sample --no-cpu-baseline > $OUT/${TAG}_sample_bs64_bf16.json 2> $OUT/sample.err
python bench.py --config fwd_bwd_fp32_bs64 --no-cpu-baseline > $OUT/${TAG}_fwd_bwd_fp32_bs64.json 2> $OUT/fp32.err
# the traffic file must sit in profiles/ for bench.py to attach it: second default line with traffic filled in
cp $OUT/${TAG}_pmc_traffic.json $OUT/${TAG}_pmc_traffic_sample.json $OUT/${TAG}_pmc_traffic_fp32.json profiles/ && python bench.py --no-cpu-baseline > $OUT/${TAG}_bench_with_traffic.json 2>> $OUT/bench.err
rm -rf $OUT/prof $OUT/pmc_tcc $OUT/pmc_tcc_s $OUT/pmc_tcc_f $OUT/pmc_sq
ls -la $OUT
