"""What the compiler made of the solve kernels, read from the code objects the build produced (no GPU): the lean form
(spectral_amd/csrc/btrapz_lean_body.h) exists to run TWO wavefronts per SIMD -- at most 256 registers, none of them
accumulation registers, and 20 KB of LDS per wavefront -- and the packed form must not spill."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

from spectral_amd import native

OBJDIR = os.path.join(native.LIB_DIR, "obj")
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels_of(obj):
    if not os.path.exists(os.path.join(LLVM, "llvm-objdump")):
        pytest.skip("no llvm-objdump")
    if not os.path.exists(os.path.join(OBJDIR, obj)):
        native.build()
    with tempfile.TemporaryDirectory() as d:
        shutil.copy(os.path.join(OBJDIR, obj), d)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", obj], cwd=d, check=True, capture_output=True)
        dev = [f for f in os.listdir(d) if "amdgcn" in f]
        assert len(dev) == 1, os.listdir(d)
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", dev[0]], cwd=d, check=True, capture_output=True, text=True).stdout
    out = {}
    for block in notes.split("  - .agpr_count:")[1:]:
        g = lambda key: int(re.search(r"\.%s:\s*(\d+)" % key, block).group(1))
        name = re.search(r"\.name:\s*(\S+)", block).group(1)
        out[name] = dict(agpr=int(block.split()[0]), vgpr=g("vgpr_count"), sgpr=g("sgpr_count"), lds=g("group_segment_fixed_size"),
                         scratch=g("private_segment_fixed_size"))
    return out


def test_lean_kernels_fit_two_wavefronts_per_simd():
    ks = {}
    ks.update(kernels_of("btrapz_lean.o")); ks.update(kernels_of("btrapz_lean_warm.o"))
    lean = {n: r for n, r in ks.items() if "ipm_solve_lean" in n}
    assert len(lean) == 7, sorted(lean)   # (round 5: 11 until the ordered instantiations served ragged and uniform batches alike)
    for name, r in lean.items():
        # 512 registers per SIMD lane: two wavefronts need <= 256 each, all architectural; 160 KB of LDS per CU over eight
        assert r["vgpr"] <= 256 and r["agpr"] == 0, (name, r)
        assert r["lds"] == 20480, (name, r)
        # scratch: read-mostly problem data the allocator evicts -- a bound, so that a change that makes the
        # kernel spill its state (the 744 B of the packed form under the same budget: 12.5 ms) does not pass unnoticed
        # (round 4, final: 96-120 B in the cold instantiations, 188-212 B in the warm-start ones)
        # (round 5, final: 92-116 B in the cold instantiations, 188-196 B in the warm-start ones)
        assert r["scratch"] <= (200 if "warm" in name else 120), (name, r)


def test_packed_kernels_do_not_spill():
    ks = kernels_of("btrapz_kernels.o")
    solve = {n: r for n, r in ks.items() if "ipm_solve_" in n}
    assert len(solve) >= 11      # (round 5: the queue kernel is compiled into -DBTRAPZ_EXPERIMENTS builds only)
    for name, r in solve.items():
        # (.vgpr_count of the metadata is the unified allocation: 256 architectural + the accumulation registers)
        # Cold instantiations: no scratch.  The warm-start ones sit at the full 512 registers and since round 5 (the
        # far-row pass of their in-loop cold restart, btrapz_ipm.h "bounds that are no bounds") spill 40 B: five values
        # outside the loop, one dword in it (3 reloads + 2 stores per iteration of ~10 000 instructions: read off the ISA).
        assert r["scratch"] <= (48 if "warm" in name else 0) and 256 < r["vgpr"] <= 512 and r["agpr"] > 0, (name, r)
