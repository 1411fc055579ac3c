#!/bin/bash
# GPU box: the round's measurement artifacts -> gpurun_out/final_<tag>/ (copy the summaries into profiles/ afterwards)
set -eu
TAG=${1:-e}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/final_$TAG
mkdir -p $OUT
cd $R && python bench.py > $OUT/bench_full.log 2>&1
grep -o '{"metric.*' $OUT/bench_full.log > $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --no-cpu-baseline > $OUT/stats_bench.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 4 --warmup 2 --no-graphs --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -- python3 $R/bench.py --steps 4 --warmup 2 --no-graphs --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
cd $R && python tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_traffic.json
cp $OUT/stats/*/*kernel_stats.csv $OUT/kernel_stats.csv
# keep the merge small: drop the raw traces
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write
cut -c1-400 $OUT/bench.json
# the opt-in bf16x6 arithmetic mode: bench line + kernel stats
cd $R && python bench.py --precision bf16x6 --no-cpu-baseline > $OUT/bench_bf16x6_full.log 2>&1
grep -o '{"metric.*' $OUT/bench_bf16x6_full.log > $OUT/bench_bf16x6.json
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats6 -- python3 $R/bench.py --precision bf16x6 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
cp $OUT/stats6/*/*kernel_stats.csv $OUT/kernel_stats_bf16x6.csv; rm -rf $OUT/stats6
cut -c1-200 $OUT/bench_bf16x6.json
