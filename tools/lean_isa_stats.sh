#!/bin/bash
# tools/lean_isa_stats.sh [flags]: ISA of k_base into /tmp/asm/b.s + instruction statistics
cd "$(dirname "$0")/../spectral_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -I. "$@" -S --cuda-device-only -o /tmp/asm/three.s /tmp/lean_three.hip 2>/dev/null
cd /tmp/asm; awk '/^_ZN6btrapz6k_base/,/\.Lfunc_end0/' three.s > b.s
echo "lines $(wc -l < b.s)  valu $(grep -c '^\s*v_' b.s)  f64 $(grep -c 'v_fma_f64\|v_mul_f64\|v_add_f64\|v_fmac_f64\|v_max_f64\|v_min_f64\|v_rcp_f64' b.s)  mov $(grep -c 'v_mov_b32_e\|v_mov_b64' b.s)  dpp $(grep -c _dpp b.s)  cnd $(grep -c v_cndmask b.s)  readlane $(grep -c v_readlane b.s)  writelane $(grep -c v_writelane b.s)  s_load $(grep -c s_load b.s)  scratch_ld $(grep -c scratch_load b.s) scratch_st $(grep -c scratch_store b.s) ds $(grep -c '^\s*ds_' b.s) vmem $(grep -c 'global_load\|flat_load' b.s)"
grep "codeLenInByte\|; NumSgprs\|; ScratchSize" three.s | head -3
