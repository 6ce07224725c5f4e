#!/bin/bash
# Counters of the solve kernels on one bench batch (separate --pmc passes, no tracing):
#   gpurun -- 'bash tools/lean_pmc.sh <tag> [case]'   ->  gpurun_out/<tag>/pmc*/ ; summarise with tools/summarize_lean_pmc.py
set -u
TAG=${1:-lean}
CASE=${2:-0}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
i=0
for group in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
  "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
  "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SMEM SQ_INSTS_VALU_INT32" \
  "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE" ; do
  i=$((i + 1))
  # shellcheck disable=SC2086
  rocprofv3 --pmc $group --output-format csv -d "$OUT/pmc$i" -- python3 "$ROOT/tools/lean_bench.py" --cases "$CASE" --reps 1 --oracle 0 \
    > "$OUT/pmc$i.json" 2> "$OUT/pmc$i.err" || echo "FAILED pmc$i" >> "$OUT/failed.txt"
done
find "$OUT" -name "*.db" -delete 2>/dev/null
du -sh "$OUT"
