"""N > 1 path on CPU: world_size-2 gloo.  Shards are independent; the only exchange is the
(cost, index) all_gather + local min of spectral_amd/dist.py (bench.py uses the same code)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from spectral_amd.dist import fetch_winner, global_argmin, global_argmin_with_winner, shard_bounds


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, costs, groups, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B = costs.shape[0]
    lo, hi = shard_bounds(B, world, rank)
    local = torch.from_numpy(costs[lo:hi])
    n = hi - lo
    per = n // groups
    bc = torch.empty(groups, dtype=torch.float64); bi = torch.empty(groups, dtype=torch.int64)
    for g in range(groups):  # local arg-min per group, ties -> lowest index, none solved -> -1
        seg = local[g * per:(g + 1) * per]
        j = int(torch.argmin(seg))
        bc[g] = seg[j]; bi[g] = lo + g * per + j if torch.isfinite(seg[j]) else -1
    c, i = global_argmin(bc, bi)
    q.put((rank, c.numpy().copy(), i.numpy().copy()))
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_global_argmin_equals_single_process(world):
    rng = np.random.default_rng(5)
    B, groups = 4096, 4
    costs = rng.normal(size=B) * 1e4
    costs[rng.integers(0, B, 300)] = np.inf                 # failed candidates carry +inf
    costs[100] = costs[3000] = costs.min() - 1.0             # a tie across ranks -> lowest index wins
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, costs, groups, q)) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=120) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    per = B // world // groups
    for rank, c, i in res:
        for g in range(groups):
            # group g of the global view = union over ranks of their g-th local group
            cand = np.concatenate([np.arange(r * (B // world) + g * per, r * (B // world) + (g + 1) * per) for r in range(world)])
            j = cand[np.argmin(costs[cand])]
            assert i[g] == j and c[g] == costs[j]
    assert all((res[0][2] == r[2]).all() for r in res)       # every rank agrees
    assert 100 in res[0][2]


def test_shard_bounds_cover_the_batch():
    for B, W in [(65536, 8), (4096, 3), (5, 8), (1, 1)]:
        cover = []
        for r in range(W):
            lo, hi = shard_bounds(B, W, r)
            cover += list(range(lo, hi))
        assert cover == list(range(B))


def test_the_c_abi_shards_as_dist_does():
    """btrapz_multi_shard_bounds (the one-process multi-GPU step of include/btrapz_hip.h) cuts a batch exactly where
    spectral_amd.dist.shard_bounds does; with arg-min groups a shard is a whole number of groups.  No GPU needed."""
    from spectral_amd import native
    for B, G in ((65536, 8), (1000, 3), (5, 8), (1, 1), (4097, 4), (64, 64)):
        for g in range(G):
            assert native.multi_shard_bounds(B, G, g) == shard_bounds(B, G, g), (B, G, g)
    B, G, group = 128 * 512, 8, 512                       # BASELINE config 5: 16 agents x 512 candidates per GPU
    assert [native.multi_shard_bounds(B, G, g, group) for g in range(G)] == [(g * 8192, (g + 1) * 8192) for g in range(G)]
    cuts = [native.multi_shard_bounds(12 * 64, 5, g, 64) for g in range(5)]        # 12 groups over 5 devices: 3, 3, 3, 3, 0
    assert cuts == [(0, 192), (192, 384), (384, 576), (576, 768), (768, 768)]
    with pytest.raises(native.BtrapzError):
        native.multi_shard_bounds(10, 2, 0, 3)            # 10 % 3 != 0
    with pytest.raises(native.BtrapzError):               # without a HIP device there is no handle, and no CPU stand-in
        if torch.cuda.is_available():
            raise native.BtrapzError("box has a GPU")
        native.MultiContext([0, 0])


def test_all_failed_group_reports_minus_one():
    c, i = global_argmin(torch.tensor([float("inf")], dtype=torch.float64), torch.tensor([-1]))
    assert int(i[0]) == -1


def _worker_edge(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    base = (1 << 60) + 12345                                   # indices beyond 2^53: exact only as integers
    # group 0: a tie between the ranks -> lowest index; group 1: rank 0 failed, rank 1 solved; group 2: nobody solved;
    # group 3: a NaN cost (never produced by the kernels, must still not win)
    bc = torch.tensor([[-7.5, float("inf"), float("inf"), float("nan")], [-7.5, 3.0, float("inf"), 5.0]][rank], dtype=torch.float64)
    bi = torch.tensor([[base + 9, -1, -1, base + 1], [base + 2, base + 77, -1, base + 3]][rank], dtype=torch.int64)
    c, i = global_argmin(bc, bi)
    q.put((rank, c.numpy().copy(), i.numpy().copy()))
    dist.barrier(); dist.destroy_process_group()


def test_global_argmin_edge_cases_and_exact_indices():
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_worker_edge, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=120) for _ in range(2)]
    [p.join(timeout=60) for p in procs]
    base = (1 << 60) + 12345
    for rank, c, i in res:
        assert i.tolist() == [base + 2, base + 77, -1, base + 3]
        assert c[0] == -7.5 and c[1] == 3.0 and np.isinf(c[2]) and c[3] == 5.0


def _worker_winner(rank, world, port, costs, ctrl, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, P = ctrl.shape
    lo, hi = shard_bounds(B, world, rank)
    per = (B + world - 1) // world
    local = torch.from_numpy(costs[lo:hi]); lctrl = torch.from_numpy(ctrl[lo:hi])
    j = int(torch.argmin(local))
    solved = bool(torch.isfinite(local[j]))
    bc = local[j:j + 1].clone(); bi = torch.tensor([lo + j if solved else -1], dtype=torch.int64)
    c, i, w = global_argmin_with_winner(bc, bi, lctrl[j:j + 1])           # one all_gather: pair + control points
    f = fetch_winner(lctrl, int(i[0]), per, lo)                            # the broadcast form
    q.put((rank, float(c[0]), int(i[0]), w[0].numpy().copy(), None if f is None else f.numpy().copy()))
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.parametrize("case", ["rank0 wins", "rank1 wins", "nobody solved", "one rank failed"])
def test_winner_control_points_reach_every_rank(case):
    """after the arg-min the winner's control points (12 S doubles) travel from their owner to every rank --
    in the same all_gather as the (cost, index) pair (global_argmin_with_winner), or by one broadcast from the owner
    (fetch_winner).  Both give the owner's row bit for bit, on every rank."""
    rng = np.random.default_rng(11)
    B, P = 1001, 240
    costs = rng.normal(size=B) * 1e3
    ctrl = rng.normal(size=(B, P))
    if case == "rank0 wins": costs[17] = -1e9
    if case == "rank1 wins": costs[900] = -1e9
    if case == "nobody solved": costs[:] = np.inf
    if case == "one rank failed": costs[:501] = np.inf
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_worker_winner, args=(r, 2, port, costs, ctrl, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=120) for _ in range(2)]
    [p.join(timeout=60) for p in procs]
    want = int(np.argmin(costs)) if np.isfinite(costs).any() else -1
    for rank, c, i, w, f in res:
        assert i == want
        if want < 0:
            assert np.isnan(w).all() and f is None and np.isinf(c)
        else:
            assert c == costs[want] and np.array_equal(w, ctrl[want]) and np.array_equal(f, ctrl[want])


def test_winner_of_a_single_process_is_its_local_winner():
    c, i, w = global_argmin_with_winner(torch.tensor([2.5], dtype=torch.float64), torch.tensor([7]), torch.arange(6, dtype=torch.float64)[None])
    assert int(i[0]) == 7 and torch.equal(w[0], torch.arange(6, dtype=torch.float64))
    assert fetch_winner(torch.arange(12, dtype=torch.float64).view(2, 6), 1, 2, 0).tolist() == [6, 7, 8, 9, 10, 11]


def test_strong_scaling_shards_are_the_one_batch():
    """bench.py --scaling strong: every rank draws the same batch and solves its contiguous shard."""
    import bench
    from spectral_amd import layout as L
    full, sh = bench.make_workload("scenario1", 1000, 20, 0, 0)
    again, _ = bench.make_workload("scenario1", 1000, 20, 0, 0)
    assert np.array_equal(full.seg, again.seg) and np.array_equal(full.init, again.init)      # deterministic
    parts = []
    for r in range(3):
        lo, hi = shard_bounds(1000, 3, r)
        parts.append(full.slice(lo, hi))
    assert sum(p.B for p in parts) == 1000
    assert np.array_equal(np.concatenate([p.seg for p in parts], axis=1), full.seg)
    assert np.array_equal(np.concatenate([p.init for p in parts]), full.init)
    other, _ = bench.make_workload("scenario1", 1000, 20, 0, 1)                               # weak: a batch per rank
    assert not np.array_equal(other.init, full.init)
    assert full.seg.shape == (L.NUM_SEG_FIELDS, 1000, 20)


def test_bench_refuses_more_ranks_than_devices():
    """python bench.py --gpus N without a torchrun environment launches the N ranks itself, and says so loudly when
    the box has fewer devices instead of benchmarking one GPU under the wrong label."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if torch.cuda.device_count() >= 2:
        pytest.skip("two devices present")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 3 and "HIP device" in p.stderr and p.stdout.strip() == ""


def test_launcher_starts_the_one_process_host_when_the_ranks_end_without_a_line():
    """`bench.py --gpus N` has a second way to produce its line.  When the torch.distributed ranks end
    non-zero before a line was printed (here: no HIP device, so every rank refuses at once; on a GPU box: RCCL's bootstrap
    or IPC handles -- tests/test_gpu_bench_ranks.py), the LAUNCHER -- which never touches the GPU -- starts
    `--host one-process` (the C-ABI's btrapz_multi_* step) as a fresh child and returns ITS exit code.  On this CPU box
    that child refuses too (there is no CPU path), loudly."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present: the GPU suite runs the fallback to a real line")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-device", "--backend", "gloo", "--steps", "1",
                        "--warmup", "0", "--batch", "600", "--no-cpu-baseline", "--latency-reps", "0", "--no-secondary"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode != 0 and p.stdout.strip() == ""
    err = p.stderr
    first = err.index("the 2 ranks ended with exit code")
    assert "running the step with --host one-process" in err[first:]
    assert err.count("bench.py needs a HIP device: the hot path has no CPU fallback") == 3     # two ranks, then the one process
    assert err.rindex("needs a HIP device") > first                                            # ... the last one AFTER the hand-over


def test_one_process_argv_drops_what_belongs_to_the_ranks(monkeypatch):
    import sys
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--backend", "nccl", "--steps=7", "--master-port", "29511", "--host", "ranks",
                                      "--scaling", "strong", "--backend=gloo", "--share-device"])
    assert bench.one_process_argv() == ["--gpus", "4", "--steps=7", "--scaling", "strong", "--share-device"]


def _launch(env_extra, timeout):
    import subprocess, sys, time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    t0 = time.time()
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(os.path.dirname(os.path.abspath(__file__)), "rank_fail_worker.py")],
                       capture_output=True, text=True, env=env, timeout=timeout)
    return p, time.time() - t0


def _rank_reports(d):
    return {f: open(os.path.join(d, f)).read() for f in sorted(os.listdir(d))}


def test_a_failing_rank_ends_the_launch(tmp_path):
    """First-contact hardening: process groups are created with a finite timeout
    (spectral_amd.dist.init_process_group); a rank that dies before its first collective makes the launcher return
    non-zero -- it ends the rank that is waiting in the all-gather -- instead of hanging.  The healthy launch of the same
    worker returns 0 and both ranks agree on the winner."""
    good_dir, bad_dir = tmp_path / "good", tmp_path / "bad"
    good_dir.mkdir(); bad_dir.mkdir()
    ok, _ = _launch({"FAIL_RANK": "-1", "RESULT_DIR": str(good_dir)}, 240)
    assert ok.returncode == 0, ok.stderr[-1500:]
    # each rank reports into its own file (the shared stdout pipe interleaves the ranks' writes: ADVICE r4)
    assert _rank_reports(str(good_dir)) == {"rank0.txt": "rank 0 winner 0\n", "rank1.txt": "rank 1 winner 0\n"}
    bad, dt = _launch({"FAIL_RANK": "1", "COLLECTIVE_TIMEOUT_S": "20", "RESULT_DIR": str(bad_dir)}, 240)
    assert bad.returncode != 0 and dt < 120, (bad.returncode, dt)
    reports = _rank_reports(str(bad_dir))
    assert reports.get("rank1.txt") == "rank 1 fails before its first collective\n"
    assert "rank0.txt" not in reports            # rank 0 never got a winner: it was ended inside the all-gather
