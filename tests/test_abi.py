"""The C-ABI boundary without a GPU: every symbol include/btrapz_hip.h declares is exported,
the three drop-in libraries export exactly `find_traj`, struct layouts match the reference's
ctypes definitions, and -- with no HIP device -- the product fails loudly instead of falling
back to a CPU path."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from spectral_amd import native, synth, trp_wrapper, cub_wrapper

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "btrapz_hip.h")


@pytest.fixture(scope="module")
def built():
    native.build()
    return native.lib()


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(btrapz_[a-z_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(built):
    names = declared_functions()
    assert "btrapz_solve_batch_device" in names and "btrapz_find_traj" in names
    for n in names:
        assert hasattr(built, n), n
    assert set(names) == set(native.EXPORTS)


@pytest.mark.parametrize("lib", ["libtrp.so", "libcub.so", "libbtrapz.so"])
def test_dropin_libraries_export_only_find_traj(built, lib):
    path = os.path.join(native.LIB_DIR, lib)
    out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
    syms = [l.split()[-1] for l in out.splitlines() if " T " in l]
    assert syms == ["find_traj"]            # nm -D of the reference's libtrp.so: `T find_traj` only


def test_params_layout_is_the_reference_abi():
    # include/btrapz/py_cpp_.h:6-21: 10 doubles + int, natural alignment -> 88 bytes
    for P in (native.CParams, trp_wrapper.Params, cub_wrapper.Params):
        assert C.sizeof(P) == 88
        assert [f[0] for f in P._fields_] == ["s_acc_weight", "s_jerk_weight", "l_acc_weight", "l_jerk_weight",
                                               "weight_s_ref", "weight_ds_ref", "weight_l_ref", "weight_dl_ref",
                                               "weight_end_s", "weight_end_l", "iteration"]
        assert P.iteration.offset == 80
    assert C.sizeof(native.CShared) == 21 * 8 + 8
    assert C.sizeof(native.CSegment) == 104
    assert C.sizeof(native.COptions) == 80               # btrapz_options: two ints, three doubles, int (+pad), two doubles, five ints (+pad)
    assert native.COptions.struct_size.offset == 0 and native.COptions.max_iter.offset == 4
    assert native.COptions.elastic.offset == 32 and native.COptions.elastic_delta.offset == 48
    o = native.COptions(); o.max_iter = 7; native.lib().btrapz_options_init(C.byref(o))
    assert (o.struct_size, o.max_iter, o.elastic, o.queue, o.split, o.start, o.cap_iter, o.lean) == (80, 0, 0, 0, 0, 0, 0, 0)
    assert native.COptions.cap_iter.offset == 68 and native.COptions.lean.offset == 72
    assert C.sizeof(native.CWarm) == 3 * 8 + 2 * 8 + 8   # btrapz_warm: three pointers, two doubles, one pointer
    assert [f[0] for f in native.CWarm._fields_] == ["x0", "lam0", "lam_out", "mu0", "smin", "hint"]
    assert C.sizeof(native.CTrajInput) == 2 * 4 + 8 + 6 * 8 + 2 * 8 + 8 * 8 + 6 * 8     # btrapz_traj_input
    assert native.CTrajInput.s_bounds.offset == 144


def test_no_gpu_means_loud_failure_not_cpu_fallback(built, tmp_path):
    if built.btrapz_device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(native.BtrapzError):
        native.Context(0)
    # find_traj: the sentinel the harness treats as failure (trp_wrapper.cpp:199), never a CPU answer
    gold = os.path.join(ROOT, "tests", "golden", "inputs")
    w = np.loadtxt(os.path.join(gold, "weights.txt"))
    cost = native.find_traj_native(0, native.CParams(*[float(v) for v in w], 3), os.path.join(gold, "c1.txt"),
                                   str(tmp_path / "o.txt"))
    assert cost == 100000000000.0
    assert not os.path.exists(str(tmp_path / "o.txt"))
    from spectral_amd import knots
    cost, traj, ctrl = native.find_traj_mem(0, native.CParams(*[float(v) for v in w], 3),
                                            knots.parse_corridor_file(os.path.join(gold, "c1.txt")))
    assert cost == 100000000000.0 and traj is None


def test_python_mirror_follows_reference_call_shapes(built, tmp_path, monkeypatch):
    """trp_wrapper.find_traj() -> bool, weights from a tab-separated row (trp_wrapper.py:99-121)."""
    gold = os.path.join(ROOT, "tests", "golden", "inputs")
    monkeypatch.setenv("BTRAPZ_WEIGHTS", os.path.join(gold, "weights.txt"))
    monkeypatch.setenv("BTRAPZ_INPUT", os.path.join(gold, "c1.txt"))
    monkeypatch.setenv("BTRAPZ_OUTPUT_PREFIX", str(tmp_path / "s1_slt_3d_"))
    assert len(trp_wrapper.read_weights()) == 10
    ok = trp_wrapper.find_traj()
    if built.btrapz_device_count() == 0:
        assert ok is False                      # no device -> sentinel -> False, as the harness expects
    else:
        assert ok is True and os.path.exists(str(tmp_path / "s1_slt_3d_3.txt"))


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under spectral_amd/ may reference it."""
    pkg = os.path.join(ROOT, "spectral_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h", ".c")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f)).read()
                assert "btrapz_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f


def test_bad_arguments_are_rejected_without_a_device(built):
    assert built.btrapz_solve_batch_device(None, None, None, 1, 1, None, None, None, None, None, None, None, None,
                                           None) == -1
    h = C.c_void_p()
    assert built.btrapz_create(C.byref(h), -1) in (-2,)


def test_public_header_is_plain_c99(tmp_path):
    """include/btrapz_hip.h is the drop-in boundary: it must compile as strict C and agree with the ctypes mirrors."""
    src = tmp_path / "hdr.c"
    src.write_text('#include <stdio.h>\n#include "btrapz_hip.h"\n'
                   'int main(void) { printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(Params), sizeof(btrapz_shared), '
                   'sizeof(btrapz_options), sizeof(btrapz_warm), sizeof(btrapz_traj_input), sizeof(btrapz_segment)); return 0; }\n')
    exe = tmp_path / "hdr"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           str(src), "-o", str(exe)])
    sizes = [int(v) for v in subprocess.check_output([str(exe)], text=True).split()]
    assert sizes == [C.sizeof(native.CParams), C.sizeof(native.CShared), C.sizeof(native.COptions), C.sizeof(native.CWarm),
                     C.sizeof(native.CTrajInput), C.sizeof(native.CSegment)]
