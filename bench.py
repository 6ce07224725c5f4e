#!/usr/bin/env python3
"""bench.py -- trajectory-QP solves/sec of the HIP hot path on N MI355X GPUs.

A "step" is one pass of the hot path over one batch of synthetic candidate corridors that
is already resident in HBM: per-candidate QP assembly + solve (one launch), the per-rank
arg-min, and -- for N > 1 -- the RCCL all-gather of the (cost, index) pairs that picks the
global winner.  Workload: BASELINE.json config 3 (batch = 65536 scenario_1-shaped corridors,
20 segments, order 5, trapezoid constraints) on every GPU (weak scaling: candidates are
independent, each rank owns its shard).

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X vector FP64 (64 FMA lanes/clk/CU x 256 CU x 2.4 GHz)
# Useful FP64 flops one lane (= one segment of one axis problem) spends per interior-point
# iteration: rows/residuals ~420, Newton matrix + projection ~520, two solves ~600,
# own pivot step of the block LDL^T and sweeps ~160 (DESIGN.md "Flop model").
FLOPS_PER_SEGMENT_ITER = 1700.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=65536, help="candidates per GPU")
    ap.add_argument("--segments", type=int, default=20)
    ap.add_argument("--variant", type=int, default=0, help="0 trapezoid (config 3), 1 cuboid (config 4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--latency-reps", type=int, default=200,
                    help="B=1 launches for the p50 latency (0 skips them, e.g. under rocprofv3 so that the "
                         "kernel's average duration is the batch launch alone)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--share-device", action="store_true",
                    help="dry run of the N > 1 flow on a 1-GPU box: every rank uses device 0 (use with --backend gloo)")
    return ap.parse_args()


def measured_traffic(B, S, variant):
    """HBM bytes per launch from the committed PMC profile of this workload (FETCH_SIZE + WRITE_SIZE,
    separate rocprofv3 --pmc passes, profiles/*_pmc_hbm.json); None when no profile matches."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        w = d.get("workload", {})
        if (w.get("batch_per_gpu", w.get("batch")), w.get("segments"), w.get("variant")) == (B, S, variant):
            return float(d["bytes_per_launch_raw"]), os.path.basename(f)
    return None, None


def measured_sq(B, S, variant):
    """Executed FP64 flops per launch and VALU-busy share from the committed SQ counter profile of this
    workload (profiles/*_pmc_sq.json, tools/collect_profiles.sh); None when no profile matches."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_sq.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        w = d.get("workload", {})
        if (w.get("batch_per_gpu"), w.get("segments"), w.get("variant")) == (B, S, variant):
            dv = d.get("derived", {})
            return dv.get("fp64_flops_per_launch_all_lanes"), dv.get("frac_of_wave_cycles:SQ_ACTIVE_INST_VALU"), os.path.basename(f)
    return None, None, None


def cpu_baseline(batch, shared, seconds):
    """The reference's algorithm (oracle OSQP port) on the host cores, bounded sample."""
    from oracle import oracle as O
    cores = len(os.sched_getaffinity(0))
    n0 = min(batch.B, 8 * cores)
    t0 = time.perf_counter(); O.batch_solve(batch, shared, 0, n0, threads=cores); dt = time.perf_counter() - t0
    rate = n0 / dt
    n = int(max(n0, min(batch.B, rate * seconds)))
    t0 = time.perf_counter(); _, _, st, it = O.batch_solve(batch, shared, 0, n, threads=cores); dt = time.perf_counter() - t0
    t1 = time.perf_counter(); O.batch_solve(batch, shared, 0, min(n, 64), threads=1); dt1 = time.perf_counter() - t1
    return {"value": n / dt, "unit": "solves/s", "cores": cores, "kind": "port",
            "sample": "first %d candidates of the same batch, oracle OSQP port (eps 1e-5, max_iter 5000), "
                      "%d threads; median ADMM iterations %d, accepted %.3f" % (n, cores, int(np.median(it)),
                                                                                 float(np.mean((st == 1) | (st == 2)))),
            "single_thread_solves_per_s": min(n, 64) / dt1,
            "single_thread_ms_per_solve": 1e3 * dt1 / min(n, 64)}


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    from spectral_amd import native, synth
    from spectral_amd.dist import global_argmin
    from spectral_amd.solver import BatchSolver

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    if not os.path.exists(native.LIB_PATH):
        native.build()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    if a.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(a.backend)
    solver = BatchSolver(local_rank)
    dev = solver.device

    B, S = a.batch, a.segments
    config = 3 if a.variant == 0 else 4
    batch, shared = synth.make_batch(B, S, config=config, variant=a.variant, seed=synth.SEED_BASE + config + 1000 * rank)
    db = solver.upload(batch)
    index_base = rank * B

    def step():
        o = solver.solve(db, shared)                                   # assembly + solve: one launch
        bi, bc = solver.argmin(o["cost"], index_base=index_base)       # winner of this rank's shard
        wc, wi = global_argmin(bc, bi)                                  # N > 1: 16 B per rank over RCCL
        return o, wi[0], wc[0]

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(a.warmup):
        step()
    sync()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    t0 = time.perf_counter()
    for i in range(a.steps):
        ev[i][0].record()
        o = solver.solve(db, shared)
        ev[i][1].record()
        bi, bc = solver.argmin(o["cost"], index_base=index_base)
        wc, wi = global_argmin(bc, bi)
        win_idx, win_cost = wi[0], wc[0]
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kernel_ms = float(np.mean([s.elapsed_time(e) for s, e in ev]))

    status = o["status"].cpu().numpy(); iters = o["iters"].cpu().numpy()
    solved = float(np.mean((status == 1) | (status == 2)))
    mean_iters = float(np.mean(iters)) + 1.0  # iters holds the index of the last iteration

    out = None
    if rank == 0:
        total = world * B * a.steps
        value = total / elapsed
        alg_bytes = batch.algorithmic_bytes() * B
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        flops = 2.0 * B * S * mean_iters * FLOPS_PER_SEGMENT_ITER
        traffic, traffic_src = measured_traffic(B, S, a.variant)
        sq_flops, valu_busy, sq_src = measured_sq(B, S, a.variant)
        out = {
            "metric": "trajectory QP solves/sec (20-seg order-5 corridor)", "value": value, "unit": "solves/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE.json config %d: batch=%d scenario_1-shaped corridors per GPU, %d segments, "
                                   "order 5, %s constraints, arg-min over all candidates" %
                                   (config, B, S, "trapezoid-prism" if a.variant == 0 else "cuboid"),
                       "batch_per_gpu": B, "segments": S, "variant": a.variant, "parallelism": "shard%d" % world,
                       "collective": "none" if world == 1 else "%s all_gather of (cost,index), 16 B per rank" % ("rccl" if a.backend == "nccl" else a.backend)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "btrapz::ipm_solve_kernel", "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_solve": batch.algorithmic_bytes(),
                         "note": "on-chip solve: the binding resource is FP64 VALU issue + dependent sweeps, not HBM "
                                 "(SURVEY 8d); see fp64_valu",
                         "fp64_valu": {"achieved_tflops": flops / (kernel_ms * 1e-3) / 1e12, "peak_tflops": FP64_VALU_PEAK_TFLOPS,
                                       "frac": flops / (kernel_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                                       "model": "%.0f useful flops per segment per iteration x %.2f mean iterations" %
                                                (FLOPS_PER_SEGMENT_ITER, mean_iters),
                                       # what the SIMDs actually executed (SQ_INSTS_VALU_{FMA,MUL,ADD,TRANS}_F64 x 64
                                       # lanes, FMA = 2), from the committed counter profile of this workload
                                       "executed_tflops": None if sq_flops is None else sq_flops / (kernel_ms * 1e-3) / 1e12,
                                       "executed_frac": None if sq_flops is None else sq_flops / (kernel_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                                       "valu_busy_frac_of_wave_cycles": valu_busy, "counter_source": sq_src}},
            "solved_fraction": solved, "mean_ipm_iterations": mean_iters,
            "winner": {"index": int(win_idx.item()), "cost": float(win_cost.item())},
        }
        # p50 latency of ONE solve (B = 1), inputs resident, including the sync
        if a.latency_reps > 0:
            one = solver.upload(batch.slice(0, 1))
            lat = []
            for i in range(a.latency_reps + 20):
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                solver.solve(one, shared)
                torch.cuda.synchronize(dev)
                lat.append(time.perf_counter() - t1)
            lat = np.array(lat[20:]) * 1e3
            out["p50_solve_latency_ms"] = float(np.percentile(lat, 50))
            out["p99_solve_latency_ms"] = float(np.percentile(lat, 99))
            lat_h = []
            b1 = batch.slice(0, 1)
            for i in range(max(20, a.latency_reps // 4)):
                t1 = time.perf_counter(); solver.ctx.solve_host(b1, shared); lat_h.append(time.perf_counter() - t1)
            out["p50_solve_latency_with_pcie_ms"] = float(np.percentile(np.array(lat_h[5:]) * 1e3, 50))
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(batch, shared, a.cpu_seconds)
        else:
            out["cpu_baseline"] = None
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
