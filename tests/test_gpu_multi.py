"""The multi-GPU step behind the C-ABI (include/btrapz_hip.h, btrapz_multi_*; spectral_amd/csrc/btrapz_multi.hip): one
host process, a context + stream per device slot, contiguous shards, per-device solve + local arg-min, ONE gather of
G x (16 + 96 S) bytes, the same lexicographic min on every device.  On the driver's one-GPU box the slots are LOGICAL
devices (the ordinal 0 repeated): the whole code path -- sharding, per-slot streams and buffers, pack, gather (stream-
ordered copies), select -- runs, and must return the single-context result bit for bit.  RCCL itself is touched at one
rank: ncclCommInitAll + ncclAllGather inside ncclGroupStart/End through the library's dlopen'ed librccl.so, and
torch.distributed's nccl backend at world size 1.  The reference has no counterpart (one corridor per call inside
src/cart_frenet.py:1516-1571)."""
import os

import numpy as np
import pytest

from spectral_amd import native, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def single():
    """The one-context answer the sharded step must reproduce: (batch, shared, results) per workload."""
    import torch
    from spectral_amd.solver import BatchSolver
    solver = BatchSolver(0)
    out = {}
    for key, (batch, sh) in {"s1": synth.make_scenario1_batch(3000, 20, 0), "cub": synth.make_scenario1_batch(1500, 20, 1),
                             "c2": synth.make_batch(1000, 10, config=2)}.items():
        res = {}
        for form, kw in (("lean", dict(lean=1, cap_iter=-1, split=-1)), ("packed", dict(lean=-1, cap_iter=-1, split=-1))):
            o = solver.solve(solver.upload(batch), sh, **kw)
            torch.cuda.synchronize()
            res[form] = {k: v.cpu().numpy().copy() for k, v in o.items()}
        out[key] = (batch, sh, res)
    return out


def winner_of(cost):
    c = np.where(np.isfinite(cost), cost, np.inf)
    i = int(np.argmin(c))
    return (i, float(c[i])) if np.isfinite(c[i]) else (-1, float("inf"))


@pytest.mark.parametrize("G", [1, 2, 3, 4, 7])
@pytest.mark.parametrize("key,form", [("s1", "lean"), ("s1", "packed"), ("cub", "lean"), ("c2", "packed")])
def test_logical_devices_return_the_single_context_result_bit_for_bit(single, G, key, form):
    batch, sh, res = single[key]
    want = res[form]
    m = native.MultiContext([0] * G, native.MULTI_COPIES)
    assert m.transport() == native.MULTI_COPIES
    m.upload(batch)
    kw = dict(lean=1, cap_iter=-1, split=-1) if form == "lean" else dict(lean=-1, cap_iter=-1, split=-1)
    for _ in range(2):        # (a second step on the same handle: the records of the first are overwritten in stream order)
        m.solve_argmin(sh, **kw)
    got = m.download()
    assert np.array_equal(got["status"], want["status"]) and np.array_equal(got["iters"], want["iters"])
    assert np.array_equal(got["cost"], want["cost"], equal_nan=True)
    ok = want["status"] > 0
    assert ok.any() and np.array_equal(got["ctrl"][ok], want["ctrl"][ok])
    wi, wc = winner_of(want["cost"])
    for slot in [-1] + list(range(G)):          # every device ends with the same winner
        bi, bc, bx = m.result(slot)
        assert (bi, bc) == (wi, wc), (slot, bi, bc, wi, wc)
        assert np.array_equal(bx, want["ctrl"][wi])
    # the shards are the contiguous ones of spectral_amd.dist.shard_bounds
    from spectral_amd.dist import shard_bounds
    for g in range(G):
        v = m.view(g)
        lo, hi = shard_bounds(batch.B, G, g)
        assert (v.B, v.index_base) == (hi - lo, lo) and native.multi_shard_bounds(batch.B, G, g) == (lo, hi)
    m.close()


def test_more_devices_than_candidates_and_nobody_solved(single):
    """Empty shards take part in the gather with (+inf, -1); a batch without a single solvable candidate reports -1."""
    batch, sh, _ = single["c2"]
    small = batch.slice(0, 3)
    m = native.MultiContext([0] * 5, native.MULTI_COPIES)
    m.upload(small)
    m.solve_argmin(sh, lean=-1, split=-1)
    got = m.download()
    bi, bc, bx = m.result()
    assert (got["status"] > 0).all() and bi == int(np.argmin(got["cost"])) and bc == got["cost"].min()
    assert np.array_equal(bx, got["ctrl"][bi])
    assert [m.view(g).B for g in range(5)] == [1, 1, 1, 0, 0]
    import copy
    bad = copy.copy(small); bad.init = small.init.copy(); bad.init[:, 0] = 1e6       # initial state far outside segment 0
    m.upload(bad)
    m.solve_argmin(sh, lean=-1, split=-1)
    bi, bc, bx = m.result()
    assert bi == -1 and bc == float("inf") and np.isnan(bx).all()
    m.close()


def test_groups_that_live_on_one_device_need_no_collective(single):
    """BASELINE config 5 sharded by agent: `group` candidates per ego agent, shards are whole agents, every agent's
    winner is its device's local one -- same winners as btrapz_argmin_device on one context."""
    import torch
    from spectral_amd.solver import BatchSolver
    agents, per = 12, 64
    batch, sh = synth.make_batch(agents * per, 10, config=5, agents=agents)
    solver = BatchSolver(0)
    o = solver.solve(solver.upload(batch), sh, lean=-1, split=-1, cap_iter=-1)
    wi, wc = solver.argmin(o["cost"], group=per)
    torch.cuda.synchronize()
    wi, wc, ctrl = wi.cpu().numpy(), wc.cpu().numpy(), o["ctrl"].cpu().numpy()
    for G in (1, 3, 5):
        m = native.MultiContext([0] * G, native.MULTI_COPIES)
        m.upload(batch, group=per)
        m.solve_argmin(sh, lean=-1, split=-1, cap_iter=-1)
        bi, bc, bx = m.result()
        assert np.array_equal(bi, wi) and np.array_equal(bc, wc)
        assert np.array_equal(bx, ctrl[wi])
        sizes = [m.view(g).B for g in range(G)]
        assert sum(sizes) == agents * per and all(s % per == 0 for s in sizes)
        m.close()


def test_shards_that_are_on_the_device_already(single):
    """btrapz_multi_set_shards: the caller's device arrays (e.g. the corridor stage's output), no upload."""
    import torch
    batch, sh, res = single["c2"]
    want = res["packed"]
    G = 3
    dev = torch.device("cuda", 0)
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    shards = []
    for g in range(G):
        lo, hi = native.multi_shard_bounds(batch.B, G, g)
        part = batch.slice(lo, hi)
        shards.append((hi - lo, lo, f(part.seg), f(part.init), f(part.ref_end), f(part.dl_bounds)))
    torch.cuda.synchronize()
    m = native.MultiContext([0] * G, native.MULTI_COPIES)
    m.set_shards(batch.B, batch.S, shards)
    m.solve_argmin(sh, lean=-1, cap_iter=-1, split=-1)
    got = m.download()
    assert np.array_equal(got["cost"], want["cost"], equal_nan=True)
    wi, wc = winner_of(want["cost"])
    assert m.result()[:2] == (wi, wc)
    # shards out of order / with a hole are refused
    bad = [shards[1], shards[0], shards[2]]
    with pytest.raises(native.BtrapzError):
        m.set_shards(batch.B, batch.S, bad)
    m.close()


def test_rccl_carries_the_gather_at_one_rank(single):
    """First contact with RCCL on this pool through the C-ABI: librccl.so resolved by dlopen next to the HIP runtime in
    use, ncclCommInitAll for the one device, ncclAllGather of the winner's record inside ncclGroupStart / ncclGroupEnd on
    the slot's stream -- and the result of the copy transport, bit for bit."""
    batch, sh, res = single["s1"]
    want = res["lean"]
    try:
        m = native.MultiContext([0], native.MULTI_RCCL)
    except native.BtrapzError as e:
        pytest.fail("RCCL could not be brought up at one rank: %s" % e)
    assert m.transport() == native.MULTI_RCCL and "rccl" in m.transport_library()
    m.upload(batch)
    for _ in range(3):
        m.solve_argmin(sh, lean=1, cap_iter=-1, split=-1)
    wi, wc = winner_of(want["cost"])
    bi, bc, bx = m.result()
    assert (bi, bc) == (wi, wc) and np.array_equal(bx, want["ctrl"][wi])
    m.close()
    # automatic transport with a repeated ordinal falls back to copies and says why
    m2 = native.MultiContext([0, 0], native.MULTI_AUTO)
    assert m2.transport() == native.MULTI_COPIES and "logical" in m2.fallback_reason()
    m2.close()


def test_torch_nccl_backend_at_world_size_one(tmp_path):
    """spectral_amd.dist over the REAL nccl (= RCCL) backend at world size 1, in a child process (a process group per
    interpreter): init_process_group binds the communicator to the device, all_gather_into_tensor runs on device tensors,
    global_argmin_with_winner returns the local winner through the collective path (force_collective)."""
    import subprocess
    import sys
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    code = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, %r)
from spectral_amd import dist as D, native
import torch.distributed as td
torch.cuda.set_device(0)
D.init_process_group("nccl", local_rank=0, timeout_s=60)
assert td.get_backend() == "nccl" and td.get_world_size() == 1
dev = torch.device("cuda", 0)
cost = torch.tensor([3.25, float("inf")], dtype=torch.float64, device=dev)
idx = torch.tensor([17, -1], dtype=torch.int64, device=dev)
ctrl = torch.arange(2 * 24, dtype=torch.float64, device=dev).reshape(2, 24)
ctx = native.Context(0)
c, i, x = D.global_argmin_with_winner(cost, idx, ctrl, ctx=ctx, force_collective=True)
torch.cuda.synchronize()
assert c.tolist() == [3.25, float("inf")] and i.tolist() == [17, -1]
assert torch.equal(x[0], ctrl[0]) and torch.isnan(x[1]).all()
c2, i2 = D.global_argmin(cost, idx, ctx=ctx, force_collective=True)
torch.cuda.synchronize()
assert c2.tolist() == [3.25, float("inf")] and i2.tolist() == [17, -1]
td.destroy_process_group()
print("nccl world 1 ok")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0 and "nccl world 1 ok" in p.stdout, (p.stdout[-500:], p.stderr[-2000:])


def test_a_plain_c_host_program_over_the_c_abi():
    """spectral_amd/csrc/host_check/multi_host_demo.c: no Python, no torch in the process -- gcc -std=c99, linked against
    libbtrapz_hip.so and the system HIP runtime like the drop-in libraries.  One device, three logical devices (copies) and
    RCCL at one rank: the same winner, cost and control points, printed with 17 digits."""
    import subprocess
    exe = os.path.join(native.LIB_DIR, "multi_host_demo")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", native.CSRC_DIR, "multi_demo"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    outs = []
    for args in (["1", "0"], ["1", "0", "0", "0"], ["2", "0"], ["0", "0", "0"]):
        p = subprocess.run([exe, "3000", "12"] + args, capture_output=True, text=True, env=env, timeout=300)
        assert p.returncode == 0, (args, p.stdout, p.stderr[-800:])
        line = [l for l in p.stdout.splitlines() if l.startswith("winner ")]      # (RCCL prints its version banner to stdout)
        assert len(line) == 1, p.stdout
        outs.append(line[0].split())
    f = lambda o: (o[1], o[3], o[7], o[9])          # winner, cost, two control points
    assert f(outs[0]) == f(outs[1]) == f(outs[2]) == f(outs[3]) and int(outs[0][1]) >= 0
    assert [o[5] for o in outs] == ["1", "1", "2", "1"]                 # transports: copies, copies, RCCL, auto -> copies (an ordinal repeats)


def test_argument_errors():
    with pytest.raises(native.BtrapzError):
        native.MultiContext([5])                       # no such device on this box
    with pytest.raises(native.BtrapzError):
        native.MultiContext([])
    m = native.MultiContext([0, 0], native.MULTI_COPIES)
    with pytest.raises(native.BtrapzError):
        m.solve_argmin(synth.shared_params(0))          # no batch yet
    batch, sh = synth.make_batch(10, 10, config=2)
    with pytest.raises(native.BtrapzError):
        m.upload(batch, group=3)                        # 10 % 3 != 0
    m.close()
