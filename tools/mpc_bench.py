#!/usr/bin/env python3
"""BASELINE.json config 5: receding-horizon replanning, 128 ego agents x 512 candidate corridors of 20 segments,
per-agent arg-min, cold vs warm-started solves.

    python tools/mpc_bench.py [--steps 250] [--dt 0.02] [--agents 128] [--cand 512] [--cold] [--check 3] [--gpus N]

--gpus N (SURVEY 8e, config 5 on 8 GPUs): the fleet is sharded BY AGENT -- rank r replans the agents
shard_bounds(agents, N, r) with all their candidates, so every per-agent arg-min is local to a GPU and the step needs
no collective at all; the ranks only meet after the last step to put their timings together (rank 0 reports the
fleet's rate = the slowest rank's).  Without a torchrun environment the tool launches its N ranks itself, like bench.py.

World: every candidate has a fixed corridor timeline of 1-s pieces (spectral_amd.synth, agents mode).  At step n
the horizon starts at now = n*dt: the first segment is the rest of the piece containing `now` (its lines
re-based to `now`; when less than one knot of it is left the window rolls and the next piece is extended
backwards instead, so min_first <= t_0 <= 1 + min_first, --min-first 0.1 = one knot), followed by 19 whole pieces.  Every agent's initial state is its
previous winner's state dt later (btrapz_eval_states_device); the warm start is the candidate's own previous
trajectory at the new joint times plus its previous multipliers -- except at the steps that roll the window, where
the first segment becomes another piece of the corridor and a cold start is faster (--roll: measured 7.4 ms cold,
8.0 ms from the plan alone, 13.8 ms with the multipliers rolled by one segment) --
and the previous iteration counts go in as scheduling hint (btrapz_warm.hint).

What is timed per step (HIP events around the whole step, inputs resident): window assembly (torch slicing --
workload generation, not part of the library), btrapz_eval_states_device x2, btrapz_solve_warm_device,
btrapz_argmin_device(group = candidates per agent).  One JSON line on stdout.  --dump writes the winners of a few
agents at --check steps (problem + control points) so that a test can hold them against the oracle."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(argv=None):
    """Runs the tool; returns the dict it prints as ONE JSON line (bench.py calls it in-process)."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=250)
    ap.add_argument("--dt", type=float, default=0.02, help="replanning period in seconds (50 Hz)")
    ap.add_argument("--agents", type=int, default=128)
    ap.add_argument("--cand", type=int, default=512)
    ap.add_argument("--segments", type=int, default=20)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--cold", action="store_true", help="cold start at every step (the reference's behaviour)")
    ap.add_argument("--check", type=int, default=3, help="steps whose winners are written to --dump")
    ap.add_argument("--dump", default="", help=".npz path: winners' problems and control points at the checked steps")
    ap.add_argument("--min-first", type=float, default=0.1,
                    help="shortest first segment in seconds before the window rolls (0.1 = one knot)")
    ap.add_argument("--mu0", type=float, default=0.0, help="btrapz_warm.mu0 (0 = library default)")
    ap.add_argument("--smin", type=float, default=0.0, help="btrapz_warm.smin (0 = library default)")
    ap.add_argument("--no-hint", action="store_true", help="do not pass the previous iteration counts as scheduling hint")
    ap.add_argument("--roll", default="cold", choices=["lam", "x0", "cold"],
                    help="what a step that rolls the window starts from: previous plan + rolled multipliers, plan only, nothing")
    ap.add_argument("--trace", action="store_true", help="per-step latency / iterations on stderr")
    ap.add_argument("--gpus", type=int, default=1, help="ranks = GPUs; the agents are sharded over them")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend of the N > 1 run (only the final timing gather uses it)")
    ap.add_argument("--share-device", action="store_true", help="dry run on a 1-GPU box: every rank uses device 0 (with --backend gloo)")
    ap.add_argument("--master-port", type=int, default=0)
    a = ap.parse_args(argv)

    world_env = os.environ.get("WORLD_SIZE")
    if a.gpus > 1 and world_env is None:
        # become the launcher, before anything touches the GPU (bench.py does the same)
        import socket
        import subprocess
        import torch
        if torch.cuda.device_count() < a.gpus and not a.share_device:
            sys.stderr.write("mpc_bench.py: --gpus %d but only %d HIP device(s) visible\n" % (a.gpus, torch.cuda.device_count()))
            raise SystemExit(3)
        port = a.master_port
        if not port:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
                                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)]
                                         + (list(argv) if argv is not None else sys.argv[1:]), env=env))
    world_n = int(world_env or "1") if a.gpus > 1 else 1
    rank = int(os.environ.get("RANK", "0")) if world_n > 1 else 0
    local_rank = 0 if (a.share_device or world_n == 1) else int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world_n != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world_n))

    import torch
    import torch.distributed as dist
    from spectral_amd import layout as L
    from spectral_amd import synth
    from spectral_amd.dist import shard_bounds
    from spectral_amd.solver import BatchSolver, DeviceBatch

    own_pg = False
    if world_n > 1:
        torch.cuda.set_device(local_rank)
        if not dist.is_initialized():        # (bench.py calls this tool with its process group up)
            sys.stdout.flush()
            saved_fd = os.dup(1); os.dup2(2, 1)     # (gloo announces its connections on the C-level stdout)
            try:
                from spectral_amd.dist import init_process_group
                init_process_group(a.backend, local_rank)
            finally:
                os.dup2(saved_fd, 1); os.close(saved_fd)
            own_pg = True
    S, G_all, C = a.segments, a.agents, a.cand
    horizon_pieces = S + int(np.ceil(a.steps * a.dt)) + 2
    assert horizon_pieces <= 64
    # the whole fleet's world from one seed (what a 1-GPU run sees), then this rank's agents
    world, sh = synth.make_batch(G_all * C, horizon_pieces, config=5, variant=a.variant, agents=G_all, lateral_per_agent=True)
    g_lo, g_hi = shard_bounds(G_all, world_n, rank)
    G = g_hi - g_lo
    if world_n > 1:
        world = world.slice(g_lo * C, g_hi * C)
    B = G * C
    if G < 1:
        raise SystemExit("rank %d has no agents (%d agents over %d ranks)" % (rank, G_all, world_n))
    solver = BatchSolver(local_rank)
    dev = solver.device
    wseg = torch.from_numpy(world.seg).to(dev)                       # [F][B][pieces]
    dl_bounds = torch.from_numpy(world.dl_bounds).to(dev)
    init = torch.from_numpy(world.init).to(dev).clone()
    pairs = ((L.F_DOWN_BIAS, L.F_DOWN_SKEW), (L.F_UPP_BIAS, L.F_UPP_SKEW), (L.F_L_DOWN_BIAS, L.F_L_DOWN_SKEW),
             (L.F_L_UPP_BIAS, L.F_L_UPP_SKEW), (L.F_X_BIAS, L.F_X_SKEW), (L.F_Y_BIAS, L.F_Y_SKEW))

    def window(now):
        j0 = int(np.floor(now + 1e-12))
        if j0 + 1 - now < a.min_first - 1e-12:
            j0 += 1
        off = now - j0                                               # in [-min_first, 1 - min_first]
        seg = wseg[:, :, j0:j0 + S].contiguous()
        seg[L.F_T, :, 0] = 1.0 - off
        for bias, skew in pairs:
            seg[bias, :, 0] += seg[skew, :, 0] * off
        ref_end = torch.stack([wseg[L.F_X_BIAS, :, j0 + S], wseg[L.F_Y_BIAS, :, j0 + S]], dim=1).contiguous()
        return j0, seg, ref_end

    class Win:                                                       # the DeviceBatch interface BatchSolver uses
        pass

    prev = None
    lat, iters_mean, solved = [], [], []
    dumped = []
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    check_steps = set(np.linspace(0, a.steps - 1, a.check).astype(int).tolist()) if a.check > 0 else set()
    torch.cuda.synchronize()
    wall0 = time.perf_counter()
    for n in range(a.steps):
        now = n * a.dt
        ev0.record()
        j0, seg, ref_end = window(now)
        db = Win(); db.B, db.S, db.seg, db.init, db.ref_end, db.dl_bounds = B, S, seg, init, ref_end, dl_bounds
        warm = None
        if prev is not None and not a.cold:
            times = (torch.cumsum(seg[L.F_T], dim=1) + a.dt).contiguous()
            x0 = solver.eval_states(prev["db"], prev["ctrl"], times)
            lam = prev["lam"]
            rolled = j0 != prev["j0"]
            if rolled and a.roll == "lam":                           # the window rolled: segment k was k+1
                lam = torch.roll(lam, shifts=-1, dims=3); lam[:, :, :, -1] = 0.0
            warm = dict(x0=x0, lam=lam, mu0=a.mu0, smin=a.smin)
            if rolled and a.roll == "x0":
                warm = dict(x0=x0, mu0=a.mu0, smin=a.smin)
            if rolled and a.roll == "cold":
                warm = dict()
            if not a.no_hint:
                # difficulty persists: hard / infeasible candidates share wavefronts.  Coarse classes -- everything that
                # took fewer than 8 iterations is "easy" -- so that easy candidates keep their memory order.
                warm["hint"] = torch.clamp((prev["iters"] - 4) // 4 + 1, min=1).to(torch.int32)
        out = solver.solve(db, sh, warm=warm, keep_multipliers=not a.cold,
                           out=dict(ctrl=torch.empty((B, 12 * S), dtype=torch.float64, device=dev),
                                    cost=torch.empty(B, dtype=torch.float64, device=dev),
                                    status=torch.empty(B, dtype=torch.int32, device=dev),
                                    iters=torch.empty(B, dtype=torch.int32, device=dev)))
        bi, bc = solver.argmin(out["cost"], group=C)
        # next initial state of every agent: its winner's state dt later
        nxt = solver.eval_states(db, out["ctrl"], torch.full((B, 1), a.dt, dtype=torch.float64, device=dev))
        w = bi.clamp(min=0)
        st6 = torch.cat([nxt[w, 0, 0], nxt[w, 1, 0]], dim=1)          # [G][6]
        has = (bi >= 0)[:, None]
        keep = init.view(G, C, 6)[:, 0]
        init = torch.where(has, st6, keep).repeat_interleave(C, dim=0).contiguous()
        ev1.record()
        torch.cuda.synchronize()
        lat.append(ev0.elapsed_time(ev1))
        stt = out["status"]
        solved.append(float(((stt == 1) | (stt == 2)).double().mean().item()))
        iters_mean.append(float(out["iters"].double().mean().item()) + 1.0)
        if a.trace:
            it = out["iters"].cpu().numpy() + 1
            print("step %3d j0 %2d t0 %.2f ms %.2f iters mean %.2f p50 %d p99 %d max %d" %
                  (n, j0, float(seg[L.F_T, 0, 0].item()), lat[-1], it.mean(), np.percentile(it, 50), np.percentile(it, 99), it.max()),
                  file=sys.stderr)
        if n in check_steps and a.dump:
            # winners of three agents with their problems, for tests/test_gpu_mpc_shape.py to hold against the oracle
            # (this tool never touches oracle/)
            for g in (0, G // 2, G - 1):
                wi = int(bi[g].item())
                if wi >= 0:
                    dumped.append(dict(step=n, agent=g_lo + g, seg=seg[:, wi].cpu().numpy(), init=db.init[wi].cpu().numpy(),
                                       ref_end=ref_end[wi].cpu().numpy(), dl_bounds=world.dl_bounds[wi],
                                       ctrl=out["ctrl"][wi].cpu().numpy()))
        prev = dict(db=db, ctrl=out["ctrl"], lam=out.get("lam"), j0=j0, iters=out["iters"])
    wall = time.perf_counter() - wall0
    lat = np.array(lat)
    # steps 0 and 1 are the first uses of the cold and of the warm-start kernels (code objects load on first launch: 310 ms
    # and 65-80 ms, profiles/r06_mpc_trace.txt): a replanning loop pays them once, the steady state starts at step 2
    steady = lat[2:] if len(lat) > 3 else (lat[1:] if len(lat) > 1 else lat)
    result = {
        "workload": "BASELINE.json config 5: %d agents x %d candidates, %d segments, replanned every %.0f ms, %d steps, %s"
                    % (G_all, C, S, 1e3 * a.dt, a.steps, "cold start every step" if a.cold else "warm start"),
        "mode": "cold" if a.cold else "warm", "min_first_segment_s": a.min_first,
        "achieved_hz": 1e3 / float(steady.mean()), "target_hz": 1.0 / a.dt,
        "p50_step_ms": float(np.percentile(steady, 50)), "p99_step_ms": float(np.percentile(steady, 99)),
        "first_step_ms": float(lat[0]), "second_step_ms": float(lat[1]) if len(lat) > 1 else None, "wall_s_incl_checks": wall,
        "mean_ipm_iterations": float(np.mean(iters_mean[1:] if len(iters_mean) > 1 else iters_mean)),
        "mean_ipm_iterations_first_step": iters_mean[0],
        "solved_fraction_mean": float(np.mean(solved)), "solved_fraction_min": float(np.min(solved)),
        "candidates_per_s": B * 1e3 / float(steady.mean()),
        "dumped_winners": len(dumped),
        "n_gpus": world_n, "agents_per_gpu": G, "parallelism": "agents sharded over %d GPU(s), per-agent arg-min local, no collective" % world_n,
        "last_winners": [int(v) + g_lo * C if v >= 0 else -1 for v in bi.cpu().tolist()],   # global candidate indices, this rank's agents
    }
    if world_n > 1:
        # after the last step: the ranks' figures side by side (gloo side group: python objects, not on the data path)
        import datetime
        # (gloo announces its connections on the C-level stdout: this program's stdout carries ONE JSON line, so the
        #  descriptor points at stderr while the group is set up and used)
        sys.stdout.flush()
        saved_fd = os.dup(1); os.dup2(2, 1)
        try:
            side = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=180))   # (a rank that failed must not hang the others)
            allr = [None] * world_n
            dist.all_gather_object(allr, result, group=side)
        finally:
            os.dup2(saved_fd, 1); os.close(saved_fd)
        if rank == 0:
            result = dict(allr[0])
            result["achieved_hz"] = min(r["achieved_hz"] for r in allr)          # a fleet replans as fast as its slowest shard
            result["achieved_hz_by_rank"] = [r["achieved_hz"] for r in allr]
            result["p50_step_ms"] = max(r["p50_step_ms"] for r in allr); result["p99_step_ms"] = max(r["p99_step_ms"] for r in allr)
            result["candidates_per_s"] = sum(r["candidates_per_s"] for r in allr)
            result["mean_ipm_iterations"] = float(np.mean([r["mean_ipm_iterations"] for r in allr]))
            result["solved_fraction_mean"] = float(np.mean([r["solved_fraction_mean"] for r in allr]))
            result["solved_fraction_min"] = min(r["solved_fraction_min"] for r in allr)
            result["last_winners"] = [w for r in allr for w in r["last_winners"]]
            result["dumped_winners"] = sum(r["dumped_winners"] for r in allr)
        dist.barrier(group=side)
        if own_pg:
            dist.destroy_process_group()
    if rank != 0:
        if a.dump:
            np.savez(a.dump + ".rank%d.npz" % rank, **{"%s_%d" % (k, i): v for i, d_ in enumerate(dumped) for k, v in d_.items()}, n=len(dumped))
        return result
    print(json.dumps(result))
    if a.dump:
        np.savez(a.dump, **{"%s_%d" % (k, i): v for i, d_ in enumerate(dumped) for k, v in d_.items()}, n=len(dumped))
    return result


if __name__ == "__main__":
    main()
