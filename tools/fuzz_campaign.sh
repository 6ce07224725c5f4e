#!/bin/bash
# The fuzz drivers of tests/fuzz against the oracle, at scale, on the GPU box (through gpurun from the repo root):
#   gpurun --timeout 1200 -- 'bash tools/fuzz_campaign.sh r03'
# Logs under gpurun_out/fuzz_<tag>/; the last lines of every log are the tallies DESIGN.md section 5 quotes.
TAG=${1:-r03}
CALLS=${2:-1500}
SEED=${3:-300}
PART=${4:-all}      # all | battery (rounds 2-4) | weights (round 5: the reference's weight space) -- two gpurun calls fit the limit
OUT=gpurun_out/fuzz_$TAG
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
if [ "$PART" != "battery" ]; then
  # round 5: find_traj and every batched form over the reference's weight space (all_weights.txt + U(0,50)^10 draws +
  # degenerate rows; src/trp_wrapper.py:56-97): rescue pass on / off, two seeds of draws; 3 072-candidate slices of the four
  # bench families under 8 rows chosen for spread
  timeout -k 10 900 python tests/fuzz/weights_find_traj.py 0 200 1 > "$OUT/weights_find_traj_e1.log" 2>&1; tail -n 5 "$OUT/weights_find_traj_e1.log"
  timeout -k 10 900 python tests/fuzz/weights_find_traj.py 1 200 0 "c1,c2,c_road_s1_3,c3,c6" > "$OUT/weights_find_traj_e0.log" 2>&1; tail -n 5 "$OUT/weights_find_traj_e0.log"
  timeout -k 10 1100 python tests/fuzz/weights_batched.py 3072 8 0 > "$OUT/weights_batched.log" 2>&1; tail -n 4 "$OUT/weights_batched.log"
  [ "$PART" = "weights" ] && exit 0
fi
# find_traj (in-memory) against the oracle: plain solve, then the product's default (rescue pass on) with the OSQP
# port's decision tallied on every 10th call; four processes at a time (at most 6 may hold the GPU)
for e in 0 1; do
  for s in 1 2 3 4; do
    timeout -k 10 900 python tests/fuzz/find_traj_vs_oracle.py $((SEED + 10 * e + s)) "$CALLS" $e - $((e * 10)) > "$OUT/find_traj_e${e}_s$s.log" 2>&1 &
  done
  wait
  echo "find_traj elastic=$e done"; tail -n 2 "$OUT"/find_traj_e${e}_s*.log
done
timeout -k 10 600 python tests/fuzz/prisms_vs_restatement.py 5 3000 > "$OUT/prisms.log" 2>&1; tail -n 2 "$OUT/prisms.log"
timeout -k 10 900 python tests/fuzz/warm_start_vs_oracle.py 4096 > "$OUT/warm_start.log" 2>&1; tail -n 3 "$OUT/warm_start.log"
timeout -k 10 300 python tests/fuzz/fused_vs_two_launches.py 1 200 > "$OUT/fused_corridors.log" 2>&1; tail -n 1 "$OUT/fused_corridors.log"
timeout -k 10 600 python tests/fuzz/corridors_vs_oracle.py 100 10 > "$OUT/corridors.log" 2>&1; tail -n 1 "$OUT/corridors.log"
timeout -k 10 1200 python tests/fuzz/bench_batches_vs_oracle.py 65536 16 > "$OUT/bench_batches.log" 2>&1; tail -n 6 "$OUT/bench_batches.log"
timeout -k 10 900 python tests/fuzz/lean_vs_packed.py 16384 1 > "$OUT/lean_vs_packed.log" 2>&1; tail -n 2 "$OUT/lean_vs_packed.log"
