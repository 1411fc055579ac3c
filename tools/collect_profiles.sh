#!/bin/bash
# GPU box: the round's measurement artifacts -> gpurun_out/final_<tag>/ (copy the summaries into profiles/ afterwards)
set -eu
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/final_$TAG
mkdir -p "$OUT"
# HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes) and SQ counters of the north-star launch, every arithmetic mode --
# first, and into profiles/ of this copy of the tree, so that the bench lines below carry roofline.traffic
cd "$R" && bash tools/pmc_chain.sh fp32 bf16 > "$OUT/pmc_decoder.log" 2>&1
cp gpurun_out/pmc_decoder.json "$OUT/pmc_decoder.json"; cp gpurun_out/sq_counters.json "$OUT/sq_counters.json"
cp gpurun_out/pmc_decoder.json profiles/${TAG}_pmc_decoder.json
rm -rf gpurun_out/pmcchain
cd "$R" && python bench.py > "$OUT/bench_full.log" 2>&1
grep -o '{"metric.*' "$OUT/bench_full.log" > "$OUT/bench.json"
python bench.py --precision bf16 --no-cpu-baseline > "$OUT/bench_bf16_full.log" 2>&1
grep -o '{"metric.*' "$OUT/bench_bf16_full.log" > "$OUT/bench_bf16.json"
python bench.py --precision bf16x6 --no-cpu-baseline --no-bf16-extra > "$OUT/bench_bf16x6_full.log" 2>&1
grep -o '{"metric.*' "$OUT/bench_bf16x6_full.log" > "$OUT/bench_bf16x6.json"
cd /tmp && export TMPDIR=/tmp
# rocprofv3 per-kernel durations of the same commands (graph replay); the program comes directly after `--`
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$R/bench.py" --no-cpu-baseline --no-bf16-extra > "$OUT/stats_bench.log" 2>&1
cp "$OUT"/stats/*/*kernel_stats.csv "$OUT/kernel_stats.csv"; rm -rf "$OUT/stats"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats16" -- python3 "$R/bench.py" --precision bf16 --no-cpu-baseline > "$OUT/stats16_bench.log" 2>&1
cp "$OUT"/stats16/*/*kernel_stats.csv "$OUT/kernel_stats_bf16.csv"; rm -rf "$OUT/stats16"
cd "$R"
# what each kernel family costs inside the captured step (launches of a family dropped, step re-captured and re-timed)
python tools/ablate_step.py fp32 2>&1 | grep -v amdgpu.ids > "$OUT/ablation_fp32.txt"
python tools/ablate_step.py bf16 2>&1 | grep -v amdgpu.ids > "$OUT/ablation_bf16.txt"
# BASELINE configs[1], [3], [4] at their stated size and dtype (bench.py --config)
for c in c2 c4 c5; do
  python bench.py --config $c > "$OUT/bench_${c}_full.log" 2>&1
  grep -o '{"metric.*' "$OUT/bench_${c}_full.log" > "$OUT/bench_$c.json"
done
# roofline-vs-size diagnostic: the same workload at B = 32 .. 256 clips per GPU (M = 8, T = 64), fp32 and bf16 (the chained launch
# serves B*M <= 256 workgroups; larger batches run the blocks one by one)
python tools/frac_vs_batch.py > "$OUT/frac_vs_batch.json" 2> "$OUT/frac_vs_batch.log" || true
# per-kernel timeline of one captured bf16 G-step (rocprofv3 kernel trace of a replay, tools/trace_summary.py)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr16 -- python3 "$R/tools/trace_step.py" bf16 G 12 > "$OUT/trace16.log" 2>&1
python3 "$R/tools/trace_summary.py" "$(ls /tmp/tr16/*/*kernel_trace.csv | head -1)" "$OUT/timeline_bf16_gstep.json" >> "$OUT/trace16.log" 2>&1
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr32 -- python3 "$R/tools/trace_step.py" fp32 G 12 > "$OUT/trace32.log" 2>&1
python3 "$R/tools/trace_summary.py" "$(ls /tmp/tr32/*/*kernel_trace.csv | head -1)" "$OUT/timeline_fp32_gstep.json" >> "$OUT/trace32.log" 2>&1
for K in D; do for P in fp32 bf16; do
rocprofv3 --kernel-trace --output-format csv -d /tmp/trd$P -- python3 "$R/tools/trace_step.py" $P $K 12 > /dev/null 2>&1
python3 "$R/tools/trace_summary.py" "$(ls /tmp/trd$P/*/*kernel_trace.csv | head -1)" "$OUT/timeline_${P}_dstep.json" > /dev/null 2>&1
done; done
cd "$R"
# SQ / TCC counters of every kernel of an eager fp32 G-step (the backward GEMMs: review item 1), and the residual of the captured steps
bash tools/pmc_step.sh fp32 G "$OUT/sq_backward.json" 3 > "$OUT/pmc_step.log" 2>&1 || true
bash tools/residual.sh "$OUT/residual.json" > "$OUT/residual.log" 2>&1 || true
python -m pytest tests/test_gpu_model16.py -q -m gpu > "$OUT/precision_tests.log" 2>&1; cp gpurun_out/precision_report.json "$OUT/precision_report.json"
cut -c1-300 "$OUT/bench.json"; cut -c1-300 "$OUT/bench_bf16.json"
