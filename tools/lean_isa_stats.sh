#!/bin/bash
# tools/lean_isa_stats.sh [extra hipcc flags]: static vector-instruction counts of the lean solve kernel's loop body, per
# phase (a -DLEAN_MARKS build names the phases in the ISA: A1, TERM, A2, B, C, D, E1, E2, END = what follows the loop;
# the scheduler moves arithmetic across the marks, so neighbouring phases blur into each other; the sequential loops'
# bodies are counted once).  ISA under /tmp/asm/lean_marks.s, the capped kernel alone under /tmp/asm/lean_capped.s.
set -e
cd "$(dirname "$0")/../spectral_amd/csrc"
mkdir -p /tmp/asm
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -I. -I../../include -DLEAN_MARKS "$@" \
  -S --cuda-device-only -o /tmp/asm/lean_marks.s btrapz_lean.hip 2>/dev/null
cd /tmp/asm
awk '/^_ZN6btrapz28ipm_solve_lean_capped_kernelE/,/\.Lfunc_end/' lean_marks.s > lean_capped.s
awk '/==PHASE/{ph=$NF} /^[ \t]*v_/{c[ph]++; if ($0 ~ /_dpp/) d[ph]++; if ($0 ~ /v_cndmask/) s[ph]++; if ($0 ~ /v_readlane|v_writelane/) r[ph]++; if ($0 ~ /v_mov_b64|v_mov_b32_e/) m[ph]++}
     /^[ \t]*ds_/{l[ph]++} /scratch_/{sc[ph]++}
     END{for (p in c) printf "%-5s valu %4d  dpp %3d  cndmask %3d  moves %3d  readlane/writelane %3d  lds %3d  scratch %3d\n", (p == "" ? "setup" : p), c[p], d[p], s[p], m[p], r[p], l[p], sc[p]}' lean_capped.s | sort
echo "loop body (A1 .. END mark), by opcode:"
awk '/==PHASE A1/,/==PHASE END/' lean_capped.s | grep -E "^[[:space:]]+v_" | awk '{print $1}' | sort | uniq -c | sort -rn | head -12
grep "; ScratchSize\|; NumVgprs:" lean_marks.s | sed -n 1,40p | sort | uniq -c
