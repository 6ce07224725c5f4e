#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh r02'
# then locally:  python tools/summarize_profiles.py r02
# Counter passes are separate runs with --pmc only (never combined with tracing), as the pool requires; the program
# follows `--` directly (python3 / a binary, no wrapper).  Every step's exit status is recorded: a failed step leaves a
# line in $OUT/failed.txt (and its stderr in $OUT/*.err), and tools/summarize_profiles.py refuses to summarise a
# failed or empty run.
set -u
TAG=${1:-r02}
QUICK=${2:-}                      # "quick": bench + trace + counters only; "rest": everything else (two calls fit gpurun's limit)
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
[ "$QUICK" = "rest" ] || rm -f "$OUT/failed.txt"
export TMPDIR=/tmp
BENCH="$ROOT/bench.py"
cd /tmp

step() {   # step <name> <stdout file> <command...>: run, keep stderr, record failures and empty outputs
  local name=$1 out=$2; shift 2
  if ! "$@" > "$out" 2> "$OUT/$name.err"; then echo "FAILED ($?): $name: $*" >> "$OUT/failed.txt"; return 1; fi
  if [ ! -s "$out" ]; then echo "EMPTY OUTPUT: $name: $*" >> "$OUT/failed.txt"; return 1; fi
  return 0
}

python3 -c "import sys; sys.path.insert(0, '$ROOT'); from spectral_amd import native; print(native.kernel_source_hash())" \
  > "$OUT/kernel_source_hash.txt" 2> "$OUT/hash.err" || echo "FAILED: kernel_source_hash" >> "$OUT/failed.txt"

if [ "$QUICK" != "rest" ]; then
step bench "$OUT/bench.json" python3 "$BENCH"

step trace "$OUT/bench_under_rocprof.json" rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- \
  python3 "$BENCH" --latency-reps 0 --no-cpu-baseline --no-secondary

PMC_ARGS="--steps 3 --warmup 1 --latency-reps 0 --no-cpu-baseline --no-secondary"
for group in \
  "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
  "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
  "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SMEM SQ_INSTS_VALU_INT32" \
  "GRBM_GUI_ACTIVE" ; do
  name=pmc_$(echo "$group" | tr ' ' '+' | cut -c1-60)
  # shellcheck disable=SC2086
  step "$name" "$OUT/$name.json" rocprofv3 --pmc $group --output-format csv -d "$OUT/$name" -- python3 "$BENCH" $PMC_ARGS
done

# L2 behaviour of the solve kernels (round 5: how much of the lean form's scratch re-reads and second record reads reach
# HBM).  Optional: counter names differ between rocprofv3 releases -- a failure here is noted, not fatal.
for group in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  name=pmc_opt_$(echo "$group" | tr ' ' '+' | cut -c1-60)
  # shellcheck disable=SC2086
  if ! rocprofv3 --pmc $group --output-format csv -d "$OUT/$name" -- python3 "$BENCH" $PMC_ARGS > "$OUT/$name.json" 2> "$OUT/$name.err"; then
    echo "optional counter group not collected: $group" >> "$OUT/optional_missing.txt"
  fi
done

# FETCH_SIZE / WRITE_SIZE calibration in the solve kernel's access widths (tools/fetch_calib.hip)
if /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib "$ROOT/tools/fetch_calib.hip" 2> "$OUT/calib_build.err"; then
  for mode in 8 16; do
    step "calib_fetch_$mode" "$OUT/calib_fetch_$mode.json" rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/calib_fetch_$mode" -- /tmp/fetch_calib $mode
  done
  step calib_write_48 "$OUT/calib_write_48.json" rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/calib_write_48" -- /tmp/fetch_calib 48
else
  echo "FAILED: build of tools/fetch_calib.hip" >> "$OUT/failed.txt"
fi
fi   # not "rest"

if [ "$QUICK" != "quick" ]; then
  # the other tools: BASELINE config 5 (receding horizon), knots -> control points, the other configs through bench.py
  step mpc_warm "$OUT/mpc_warm.json" python3 "$ROOT/tools/mpc_bench.py"
  step mpc_cold "$OUT/mpc_cold.json" python3 "$ROOT/tools/mpc_bench.py" --cold
  step mpc_warm_minfirst05 "$OUT/mpc_warm_minfirst05.json" python3 "$ROOT/tools/mpc_bench.py" --min-first 0.5
  step mpc_cold_minfirst05 "$OUT/mpc_cold_minfirst05.json" python3 "$ROOT/tools/mpc_bench.py" --cold --min-first 0.5
  step pipeline "$OUT/pipeline.json" python3 "$ROOT/tools/pipeline_bench.py"
  step pipeline_s1 "$OUT/pipeline_scenario1.json" python3 "$ROOT/tools/pipeline_bench.py" --scenario1
  step pipeline_prisms "$OUT/pipeline_prisms.json" python3 "$ROOT/tools/pipeline_bench.py" --prisms
  # the ragged solve in two launches (btrapz_options.cap_iter = 8; on request only: DESIGN 3.8)
  step pipeline_cap8 "$OUT/pipeline_cap8.json" python3 "$ROOT/tools/pipeline_bench.py" --cap 8
  step pipeline_s1_cap8 "$OUT/pipeline_scenario1_cap8.json" python3 "$ROOT/tools/pipeline_bench.py" --scenario1 --cap 8
  step pipeline_prisms_cap8 "$OUT/pipeline_prisms_cap8.json" python3 "$ROOT/tools/pipeline_bench.py" --prisms --cap 8
  step trace_pipeline "$OUT/pipeline_under_rocprof.json" rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_pipeline" -- \
    python3 "$ROOT/tools/pipeline_bench.py"
  step bench_config2 "$OUT/bench_config2.json" python3 "$BENCH" --segments 10 --batch 4096 --no-cpu-baseline
  step bench_config4 "$OUT/bench_config4.json" python3 "$BENCH" --variant 1 --no-cpu-baseline
  step bench_generic "$OUT/bench_generic.json" python3 "$BENCH" --workload generic --no-cpu-baseline
  step bench_2rank "$OUT/bench_2rank_gloo_strong.json" python3 "$BENCH" --gpus 2 --backend gloo --share-device --scaling strong --no-cpu-baseline --latency-reps 0
  # round 3: kernel traces of the other pipelines and of find_traj's single launch; the split form
  # against the packed one; BASELINE config 5 sharded by agent over two ranks that share the box's one GPU
  step trace_pipeline_s1 "$OUT/pipeline_scenario1_under_rocprof.json" rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_pipeline_s1" -- \
    python3 "$ROOT/tools/pipeline_bench.py" --scenario1
  step trace_pipeline_prisms "$OUT/pipeline_prisms_under_rocprof.json" rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_pipeline_prisms" -- \
    python3 "$ROOT/tools/pipeline_bench.py" --prisms
  step trace_find_traj "$OUT/find_traj_under_rocprof.json" rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_find_traj" -- \
    python3 "$ROOT/tools/find_traj_loop.py"
  export BTRAPZ_SPLIT=0
  step trace_find_traj_packed "$OUT/find_traj_packed_under_rocprof.json" rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_find_traj_packed" -- \
    python3 "$ROOT/tools/find_traj_loop.py"
  unset BTRAPZ_SPLIT
  step split_bench "$OUT/split_bench.json" python3 "$ROOT/tools/split_bench.py"
  step cap_bench "$OUT/cap_bench.json" python3 "$ROOT/tools/cap_bench.py"
  # counters of the corridor kernel (tools/summarize_corridor_pmc.py -> profiles/<tag>_corridor_pmc.json)
  i=0
  for group in \
    "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
    "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
    "SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64" ; do
    i=$((i + 1))
    # shellcheck disable=SC2086
    step "corridor_pmc$i" "$OUT/corridor_pmc$i.json" rocprofv3 --pmc $group --output-format csv -d "$OUT/corridor_pmc$i" -- python3 "$ROOT/tools/pipeline_bench.py" --reps 2
  done
  # round 6: the corridor stage alone (time, roofline fraction, a hash of every output byte) on its three workloads
  step corridor_bench "$OUT/corridor_bench.json" python3 "$ROOT/tools/corridor_bench.py"
  step corridor_bench_s1 "$OUT/corridor_bench_scenario1.json" python3 "$ROOT/tools/corridor_bench.py" --scenario1
  step corridor_bench_c1 "$OUT/corridor_bench_c1_cuboid.json" python3 "$ROOT/tools/corridor_bench.py" --input c1 --variant 1
  step mpc_2rank "$OUT/mpc_warm_2rank_gloo.json" python3 "$ROOT/tools/mpc_bench.py" --gpus 2 --backend gloo --share-device
  # round 4: the two-wavefronts-per-SIMD form against the packed one (one and two launches, small batches), its
  # hand-over sweep, the sibling warm start, and the counters of both forms' kernels on the scenario_1 batch
  step lean_bench "$OUT/lean_bench.json" python3 "$ROOT/tools/lean_bench.py" --oracle 64
  for B in 512 2048 8192 16384; do
    step lean_bench_$B "$OUT/lean_bench_B$B.json" python3 "$ROOT/tools/lean_bench.py" --batch $B --oracle 0 --reps 10 --cases 0,3
  done
  step cap_bench_lean "$OUT/cap_bench_lean.json" python3 "$ROOT/tools/cap_bench.py" --lean 1 --caps 5,6,7,8,10
  step sibling_bench "$OUT/sibling_bench.json" python3 "$ROOT/tools/sibling_bench.py"
  (cd "$ROOT" && bash tools/lean_pmc.sh "$TAG/lean_pmc" 0) > "$OUT/lean_pmc.log" 2>&1 || echo "FAILED: lean_pmc" >> "$OUT/failed.txt"
fi
# keep only the CSVs (the merge-back limit is 64 MiB)
find "$OUT" -name "*.db" -delete 2>/dev/null
du -sh "$OUT"
if [ -s "$OUT/failed.txt" ]; then echo "SOME STEPS FAILED:"; cat "$OUT/failed.txt"; exit 1; fi
echo "all steps ok"
