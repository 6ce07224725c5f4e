#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh r01'
# then locally:  python tools/summarize_profiles.py r01
# Counter passes are separate runs with --pmc only (never combined with tracing), as the pool requires.
set -u
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="$ROOT/bench.py"
cd /tmp

python3 "$BENCH" > "$OUT/bench.json" 2> "$OUT/bench.err"

rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- \
  python3 "$BENCH" --latency-reps 0 --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.err"

PMC_ARGS="--steps 3 --warmup 1 --latency-reps 0 --no-cpu-baseline"
for group in \
  "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
  "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
  "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SMEM SQ_INSTS_VALU_INT32" \
  "GRBM_GUI_ACTIVE" ; do
  name=$(echo "$group" | tr ' ' '+' | cut -c1-60)
  # shellcheck disable=SC2086
  rocprofv3 --pmc $group --output-format csv -d "$OUT/pmc_$name" -- python3 "$BENCH" $PMC_ARGS \
    > "$OUT/pmc_$name.json" 2> "$OUT/pmc_$name.err" || echo "pmc group failed: $group" >> "$OUT/failed.txt"
done
# the other tools: BASELINE config 5 (receding horizon), knots -> control points, configs 2 and 4 through bench.py
python3 "$ROOT/tools/mpc_bench.py" > "$OUT/mpc_warm.json" 2> /dev/null
python3 "$ROOT/tools/mpc_bench.py" --cold > "$OUT/mpc_cold.json" 2> /dev/null
python3 "$ROOT/tools/mpc_bench.py" --min-first 0.5 > "$OUT/mpc_warm_minfirst05.json" 2> /dev/null
python3 "$ROOT/tools/mpc_bench.py" --cold --min-first 0.5 > "$OUT/mpc_cold_minfirst05.json" 2> /dev/null
python3 "$ROOT/tools/pipeline_bench.py" > "$OUT/pipeline.json" 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_pipeline" -- \
  python3 "$ROOT/tools/pipeline_bench.py" > "$OUT/pipeline_under_rocprof.json" 2> "$OUT/trace_pipeline.err"
python3 "$BENCH" --segments 10 --batch 4096 --no-cpu-baseline > "$OUT/bench_config2.json" 2> /dev/null
python3 "$BENCH" --variant 1 --no-cpu-baseline > "$OUT/bench_config4.json" 2> /dev/null
# keep only the CSVs (the merge-back limit is 64 MiB)
find "$OUT" -name "*.db" -delete 2>/dev/null
du -sh "$OUT"
