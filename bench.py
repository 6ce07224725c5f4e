#!/usr/bin/env python3
"""bench.py -- trajectory-QP solves/sec of the HIP hot path on N MI355X GPUs.

A "step" is one pass of the hot path over one batch of synthetic candidate corridors that is already resident in HBM:
per-candidate QP assembly + solve (one launch), the per-rank arg-min, and -- for N > 1 -- the RCCL all-gather of the
(cost, index) pairs that picks the global winner.  Workload: BASELINE.json config 3 -- batch = 65536 scenario_1-shaped
corridors (src/c1.txt's corridor tiled to 20 one-second segments with per-candidate jitter, spectral_amd.synth.
make_scenario1_batch), order 5, trapezoid constraints -- per GPU (weak scaling, default) or in total (--scaling strong:
rank r owns the contiguous shard spectral_amd.dist.shard_bounds gives it).  The generic, feasible-by-construction
family of round 1 is timed beside it as a second figure.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python bench.py --gpus 8                      # spawns the 8 ranks itself (torch.distributed.run, RCCL) ...
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
         bench.py --gpus N --steps K --warmup W   # ... or runs as one of them
"""
import argparse
import json
import os
import subprocess
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
# What one solve call launches (btrapz_last_solve_form): kernel_ms in the line is the HIP-event time of the call, i.e.
# of ALL of these kernels; the rocprofv3 averages of the named ones add up to it.
SOLVE_FORMS = {0: "btrapz::ipm_solve_kernel", 1: "btrapz::ipm_solve_split_kernel", 2: "btrapz::ipm_solve_long_kernel",
               3: "btrapz::ipm_solve_capped_kernel + btrapz::ipm_solve_resume_kernel (one solve = two launches: every candidate stops when "
                  "left alone in its wavefront after 6 iterations, the second launch carries those on; + 3 bucketing launches of 5-11 us)",
               4: "btrapz::ipm_solve_queue_kernel",
               8: "btrapz::ipm_solve_lean_kernel (two wavefronts per SIMD)",
               11: "btrapz::ipm_solve_lean_capped_kernel + btrapz::ipm_solve_lean_resume_kernel (two wavefronts per SIMD; one solve = two launches: "
                   "every candidate stops when left alone in its wavefront after 6 iterations, the second launch carries those on; + 3 bucketing "
                   "launches of 5-11 us)"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=65536, help="candidates per GPU (weak) / in total (strong)")
    ap.add_argument("--segments", type=int, default=20)
    ap.add_argument("--variant", type=int, default=0, help="0 trapezoid (config 3), 1 cuboid (config 4)")
    ap.add_argument("--workload", default="scenario1", choices=["scenario1", "generic"],
                    help="scenario1: src/c1.txt's corridor tiled (BASELINE configs 3/4); generic: the random family of round 1")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the second figure on the other workload")
    ap.add_argument("--latency-reps", type=int, default=1000,
                    help="B=1 launches for the p50 latency (0 skips them, e.g. under rocprofv3 so that the "
                         "kernel's average duration is the batch launch alone)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--lean", type=int, default=0,
                    help="btrapz_options.lean: 0 the library's choice by batch size (two wavefronts per SIMD from ~3 wavefronts per SIMD "
                         "on), 1 / -1 pin the form -- the two forms agree to rounding, so a strong-scaling run that must return the "
                         "1-rank winner's control points bit for bit pins it")
    ap.add_argument("--share-device", action="store_true",
                    help="dry run of the N > 1 flow on a 1-GPU box: every rank uses device 0 (use with --backend gloo)")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port of the self-launched ranks (0: pick a free one)")
    ap.add_argument("--host", default="ranks", choices=["ranks", "one-process"],
                    help="who drives the N devices: `ranks` = one process per GPU (torch.distributed; what the driver's torchrun line "
                         "starts), `one-process` = ONE host process over the C-ABI's btrapz_multi_* step (a context + stream per device, "
                         "RCCL resolved by dlopen, else peer copies; no torch in the process).  The launcher starts the second as a fresh "
                         "child when the ranks end non-zero without a line")
    ap.add_argument("--fallback-from", default="", help=argparse.SUPPRESS)   # (set by the launcher: why the ranks did not produce the line)
    return ap.parse_args()


def self_launch(a):
    """--gpus N without a torchrun environment: become the launcher.  Runs BEFORE anything touches the GPU (counting
    devices does not initialise it); the ranks are children of torch.distributed.run, this process only waits."""
    import socket
    import torch
    have = torch.cuda.device_count()
    if have < a.gpus and not a.share_device:
        sys.stderr.write("bench.py: --gpus %d but only %d HIP device(s) visible\n" % (a.gpus, have))
        return 3
    port = a.master_port
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    # (dmabuf IPC: the only mode this pool's host driver supports -- see spectral_amd.dist.init_process_group)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rc, got_line = run_child(cmd, env)
    if rc == 0 or got_line:
        return rc                          # the launcher's exit code: non-zero when any rank failed (it ends the others)
    # The ranks ended without a line (RCCL's bootstrap, IPC handles, a rank that died): the same step has a second host
    # that needs neither torch.distributed nor IPC -- one process, all devices (btrapz_multi_*).  A FRESH child of this
    # launcher, which has not touched the GPU; never a re-exec of a process that has.
    sys.stderr.write("bench.py: the %d ranks ended with exit code %d before a line was printed: running the step with --host one-process\n" % (a.gpus, rc))
    return one_process_child(a, "torch.distributed ranks (backend %s) ended with exit code %d before their first line" % (a.backend, rc))


def run_child(cmd, env):
    """Runs cmd, passes its stdout through, says whether a JSON line was among it."""
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    got = False
    for line in p.stdout:
        got = got or line.lstrip().startswith("{")
        sys.stdout.write(line); sys.stdout.flush()
    return p.wait(), got


def one_process_argv():
    argv, skip = [], False
    for x in sys.argv[1:]:          # the same command line without what belongs to the ranks
        if skip:
            skip = False
        elif x in ("--host", "--fallback-from", "--backend", "--master-port"):
            skip = True
        elif not x.startswith(("--host=", "--fallback-from=", "--backend=", "--master-port=")):
            argv.append(x)
    return argv


def one_process_child(a, reason):
    argv = one_process_argv()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK",
                                                            "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    rc, _ = run_child([sys.executable, os.path.abspath(__file__)] + argv + ["--host", "one-process", "--fallback-from", reason], env)
    return rc


def kernel_stamp():
    from spectral_amd import native
    return native.kernel_source_hash()


def measured_profile(kind, B, S, variant, workload):
    """Counter profile of THIS kernel build on THIS workload (profiles/r*_pmc_<kind>.json, tools/collect_profiles.sh).
    A profile is used only when it carries the hash of the kernel sources it was collected with and that hash is the
    current one: numbers of an older kernel are not passed on as measurements of this one."""
    import glob
    stamp = kernel_stamp()
    stale = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_%s.json" % kind)), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        w = d.get("workload", {})
        if (w.get("batch_per_gpu", w.get("batch")), w.get("segments"), w.get("variant"), w.get("workload", "generic")) != (B, S, variant, workload):
            continue
        if d.get("kernel_source_hash") == stamp:
            return d, os.path.basename(f)
        stale = stale or os.path.basename(f)
    return None, ("stale: %s was collected with other kernel sources" % stale) if stale else None


def make_workload(name, B, S, variant, seed_offset):
    from spectral_amd import synth
    if name == "scenario1":
        return synth.make_scenario1_batch(B, S, variant, seed=synth.SEED_BASE + 3 + variant + 1000 * seed_offset)
    config = 3 if variant == 0 else 4
    return synth.make_batch(B, S, config=config, variant=variant, seed=synth.SEED_BASE + config + 1000 * seed_offset)


def workload_label(name, B, S, variant, world, scaling):
    what = ("scenario_1-shaped corridors (src/c1.txt's lane corridors, 3 m/s upper and 5 m/s lower obstacle ramps and lane "
            "change tiled to %d one-second segments; onset -1/0/+1 s, speed +-1 m/s, v0 ~ U(5,9) per candidate)" % S
            if name == "scenario1" else "generic corridors (smooth random speed profile, random margins and ramps, one lane change; "
            "feasible by construction)")
    config = 4 if variant == 1 else (2 if (B, S) == (4096, 10) else 3)
    return "BASELINE.json config %d: batch=%d %s%s, %d segments, order 5, %s constraints, arg-min over all candidates" % (
        config, B, what, " per GPU" if scaling == "weak" else " in total over %d GPU(s)" % world, S,
        "trapezoid-prism" if variant == 0 else "cuboid")


def cpu_baseline(batch, shared, seconds):
    """The reference's algorithm (oracle OSQP port at the reference's settings) on the host cores, bounded sample of
    the same batch: 1 thread, 8 threads and all schedulable cores, threads inside the C library."""
    from oracle import oracle as O
    host = O.host_cpu_info()
    cores = host["effective_cores"]
    O.fast_lib()
    t0 = time.perf_counter(); O.batch_solve(batch, shared, 0, 4, threads=1, fast=True); r1 = 4 / (time.perf_counter() - t0)
    n1 = int(max(4, min(batch.B, r1 * seconds * 0.25)))
    t0 = time.perf_counter(); _, _, st1, it1 = O.batch_solve(batch, shared, 0, n1, threads=1, fast=True); dt1 = time.perf_counter() - t0
    rates = {"1": n1 / dt1}
    n_all, dt_all, st, it = n1, dt1, st1, it1
    for th in sorted({min(8, cores), cores} - {1}):
        n = int(max(4 * th, min(batch.B, rates["1"] * th * seconds * (0.25 if th != cores else 0.5))))
        t0 = time.perf_counter(); _, _, st, it = O.batch_solve(batch, shared, 0, n, threads=th, fast=True); dt = time.perf_counter() - t0
        rates[str(th)] = n / dt
        n_all, dt_all = n, dt
    osqp = "not found"
    try:
        import ctypes
        ctypes.CDLL("libosqp.so")
        osqp = "present (not timed: its version, hence its struct layout, is unknown)"
    except OSError:
        pass
    return {"value": n_all / dt_all, "unit": "solves/s", "cores": cores, "kind": "port",
            "sample": "first %d candidates of the same batch, oracle OSQP port (eps 1e-5, max_iter 5000; gcc -O3 -march=native), "
                      "%d POSIX threads drawing candidates from a shared counter; mean ADMM iterations %.0f, accepted %.3f"
                      % (n_all, cores, float(np.mean(it)), float(np.mean((st == 1) | (st == 2)))),
            "solves_per_s_by_threads": rates,
            "scaling_efficiency_all_cores": rates[str(cores)] / (cores * rates["1"]) if cores > 1 else 1.0,
            "host": host, "single_thread_solves_per_s": rates["1"], "single_thread_ms_per_solve": 1e3 / rates["1"],
            "libosqp_so": osqp}


class HipEvents:
    """hipEvent timing on a stream of the C-ABI's own (ctypes on the HIP runtime the library is bound to; the one-process
    host has no torch in it)."""

    def __init__(self):
        import ctypes as C
        from spectral_amd import native
        native.lib()
        self.C, self.hip = C, C.CDLL(native.ROCM_HIP_RUNTIME if "torch" not in sys.modules else
                                     os.path.join(os.path.dirname(sys.modules["torch"].__file__), "lib", "libamdhip64.so"))
        self.hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        self.hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]

    def create(self):
        e = self.C.c_void_p()
        assert self.hip.hipEventCreate(self.C.byref(e)) == 0
        return e

    def record(self, e, stream):
        assert self.hip.hipEventRecord(e, stream) == 0

    def elapsed_ms(self, e0, e1):
        ms = self.C.c_float()
        assert self.hip.hipEventElapsedTime(self.C.byref(ms), e0, e1) == 0
        return float(ms.value)


def one_process_main(a):
    """--host one-process: the same step -- solve of the device's shard, local arg-min, ONE gather of (cost, index, the local
    winner's control points), the same lexicographic min on every device -- driven by ONE host process through the C-ABI
    (btrapz_multi_*, csrc/btrapz_multi.hip): a context and a stream per device, every launch asynchronous, the gather by
    RCCL (librccl resolved at run time) or, without it, by stream-ordered peer copies.  --share-device: N LOGICAL devices
    on device 0 (own context, stream, buffers and shard each; copies as transport)."""
    from spectral_amd import native
    from spectral_amd.layout import Batch
    if not os.path.exists(native.LIB_PATH):
        native.build()
    N, S = a.gpus, a.segments
    have = int(native.lib().btrapz_device_count())
    if have < 1:
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    if have < N and not a.share_device:
        sys.stderr.write("bench.py: --gpus %d but only %d HIP device(s) visible\n" % (N, have))
        return 3
    devices = [0] * N if a.share_device else list(range(N))
    if a.scaling == "weak":       # a batch per device, the ranks' batches in rank order: shard g of the C-ABI == rank g's batch
        parts = [make_workload(a.workload, a.batch, S, a.variant, r) for r in range(N)]
        shared = parts[0][1]
        cat = lambda k, ax: np.ascontiguousarray(np.concatenate([getattr(p[0], k) for p in parts], axis=ax))
        batch = Batch(B=N * a.batch, S=S, seg=cat("seg", 1), init=cat("init", 0), ref_end=cat("ref_end", 0), dl_bounds=cat("dl_bounds", 0))
        del parts
    else:
        batch, shared = make_workload(a.workload, a.batch, S, a.variant, 0)
    total = batch.B
    m = native.MultiContext(devices, native.MULTI_AUTO)
    m.upload(batch)
    call = m.prepared_step(shared, lean=a.lean)
    ev = HipEvents()
    stream0 = m.view(0).stream
    for _ in range(a.warmup):
        call()
    m.wait()
    pairs = [(ev.create(), ev.create()) for _ in range(a.steps)]
    ev.hip.hipSetDevice(devices[0])
    t0 = time.perf_counter()
    for i in range(a.steps):
        ev.record(pairs[i][0], stream0)      # device slot 0's stream: its share of the step, gather and select included
        call()
        ev.record(pairs[i][1], stream0)
    m.wait()
    elapsed = time.perf_counter() - t0
    kernel_ms = float(np.mean([ev.elapsed_ms(e0, e1) for e0, e1 in pairs]))
    win_idx, win_cost, win_ctrl = m.result()
    res = m.download()
    status, iters = res["status"], res["iters"]
    ok = (status == 1) | (status == 2)
    mean_iters = float(np.mean(iters + 1))
    B0 = m.view(0).B
    form0 = int(native.lib().btrapz_last_solve_form(m.view(0).ctx))
    rccl = m.transport() == native.MULTI_RCCL
    alg = batch.algorithmic_bytes() * B0
    achieved = alg / (kernel_ms * 1e-3) / 1e9
    flops = 2.0 * B0 * S * mean_iters * FLOPS_PER_SEGMENT_ITER
    out = {
        "metric": "trajectory QP solves/sec (20-seg order-5 corridor)", "value": total * a.steps / elapsed, "unit": "solves/s",
        "n_gpus": N, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
        "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": workload_label(a.workload, a.batch, S, a.variant, N, a.scaling), "generator": a.workload,
                   "batch_per_gpu": a.batch if a.scaling == "weak" else None, "batch_total": total, "segments": S, "variant": a.variant,
                   "parallelism": "shard%d" % N,
                   "host": "one process, %d device slot(s) %s through the C-ABI (btrapz_multi_*): a context + stream per device" % (N, devices),
                   "collective": "none" if N == 1 else (
                       "rccl ncclAllGather in one group call (%s)" % m.transport_library() if rccl else
                       "stream-ordered peer copies (hipMemcpyPeerAsync / same-device copies for logical devices)" +
                       (": " + m.fallback_reason() if m.fallback_reason() else "")) + " of (cost, index, the local winner's control points) = %d B per device and step" % (16 + 96 * S),
                   "fallback_from": a.fallback_from or None},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel": SOLVE_FORMS.get(form0, "btrapz::ipm_solve_kernel"), "kernel_ms": kernel_ms, "kernel_source_hash": kernel_stamp(),
                     "algorithmic_bytes_per_solve": batch.algorithmic_bytes(),
                     "note": "device slot 0: HIP events on ITS stream around the whole step (solve of its %d candidates + arg-min + pack + gather + "
                             "select; the solve kernels are all but ~30 us of it) -- the one-process host issues a step as ONE C call; "
                             "on-chip solve, FP64 VALU issue is the binding resource (SURVEY 8d)" % B0,
                     "fp64_valu": {"achieved_tflops": flops / (kernel_ms * 1e-3) / 1e12, "peak_tflops": FP64_VALU_PEAK_TFLOPS,
                                   "frac": flops / (kernel_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                                   "model": "%.0f useful flops per segment per iteration x %.2f mean iterations" % (FLOPS_PER_SEGMENT_ITER, mean_iters)}},
        "solved_fraction": float(ok.mean()), "mean_ipm_iterations": mean_iters,
        "status_counts": {str(int(k)): int((status == k).sum()) for k in np.unique(status)},
        "winner": {"index": int(win_idx), "cost": float(win_cost), "ctrl_sum": float(np.sum(win_ctrl)), "ctrl_head": [float(v) for v in win_ctrl[:4]]},
        "shard_sizes": [int(m.view(g).B) for g in range(N)],
        "cpu_baseline": None,
    }
    out["solved_solves_per_s"] = out["value"] * out["solved_fraction"]
    if N == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(batch, shared, a.cpu_seconds)
    m.close()
    print(json.dumps(out), flush=True)
    return 0


def main():
    a = parse()
    world_env = os.environ.get("WORLD_SIZE")
    if a.host == "one-process":
        if world_env is not None and int(os.environ.get("RANK", "0")) != 0:
            return                               # (started under torchrun: ONE of the ranks is the one process)
        raise SystemExit(one_process_main(a))
    if a.gpus > 1 and world_env is None:
        raise SystemExit(self_launch(a))
    import torch
    import torch.distributed as dist
    from spectral_amd import native
    from spectral_amd.dist import global_argmin_with_winner, shard_bounds
    from spectral_amd.solver import BatchSolver

    # This program's stdout carries ONE JSON line.  Libraries below write there too (gloo announces its connections on
    # the C-level stdout, a profiler or the runtime may): descriptor 1 points at stderr from here on and the line goes
    # to the descriptor saved now.
    sys.stdout.flush()
    line_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    world = int(world_env or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    if not os.path.exists(native.LIB_PATH):
        native.build()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    if a.share_device:
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit("rank %d: local rank %d but %d HIP device(s)" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    if world > 1:
        from spectral_amd.dist import init_process_group
        try:
            init_process_group(a.backend, local_rank)      # 60 s timeout on every collective; RCCL bound to the device at once
            # first contact, before anything is timed: one collective of the step's own kind on this rank's device
            probe = torch.full((1, 2), float(rank), dtype=torch.float64, device="cuda" if a.backend == "nccl" else "cpu")
            got = torch.empty((world, 2), dtype=torch.float64, device=probe.device)
            dist.all_gather_into_tensor(got, probe)
            torch.cuda.synchronize()
            assert got[:, 0].tolist() == [float(r) for r in range(world)], got
        except Exception as e:
            # The ranks cannot talk (RCCL's bootstrap, IPC handles, an unknown backend): the step has a second host that needs
            # neither.  Rank 0 starts it as a FRESH child process (never a re-exec: this process has initialised the GPU) and
            # leaves with its exit code; the other ranks leave quietly, so that the launcher sees the child's line and code.
            sys.stderr.write("bench.py: rank %d: first contact over %s failed: %r\n" % (rank, a.backend, e))
            try:
                dist.destroy_process_group()
            except Exception:
                pass
            if rank != 0:
                os._exit(0)
            sys.stderr.write("bench.py: rank 0: running the step with --host one-process instead\n")
            argv = [x for x in one_process_argv() if x]
            env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK",
                                                                    "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
            rc = subprocess.call([sys.executable, os.path.abspath(__file__)] + argv +
                                 ["--host", "one-process", "--fallback-from", "rank 0's first collective over %s: %s" % (a.backend, repr(e)[:160])],
                                 env=env, stdout=line_out)
            os._exit(rc)
    solver = BatchSolver(local_rank)
    dev = solver.device

    S = a.segments
    if a.scaling == "weak":
        B, index_base = a.batch, rank * a.batch
        batch, shared = make_workload(a.workload, B, S, a.variant, rank)
        total_candidates = world * B
    else:
        # strong scaling: ONE batch (every rank draws the same one), rank r solves its contiguous shard
        full, shared = make_workload(a.workload, a.batch, S, a.variant, 0)
        lo, hi = shard_bounds(a.batch, world, rank)
        batch, B, index_base = full.slice(lo, hi), hi - lo, lo
        total_candidates = a.batch
    db = solver.upload(batch)

    def winner(o):
        bi, bc = solver.argmin(o["cost"], index_base=index_base)       # winner of this rank's shard ...
        mine = o["ctrl"].index_select(0, (bi - index_base).clamp(min=0))   # ... and its control points [1][12 S]
        # N > 1: ONE all-gather of (cost, index, control points) = 16 + 96 S bytes per rank over RCCL; every rank ends
        # up with the global winner's index, cost and control points (spectral_amd/dist.py)
        return global_argmin_with_winner(bc, bi, mine, ctx=solver.ctx)

    def step():
        o = solver.solve(db, shared, lean=a.lean)                      # assembly + solve: one launch
        wc, wi, wctrl = winner(o)
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
        ev[i][0].record()                       # torch's current stream == the stream the solve is launched on
        o = solver.solve(db, shared, lean=a.lean)
        ev[i][1].record()
        wc, wi, wctrl = winner(o)
        win_idx, win_cost = wi[0], wc[0]
    sync()
    elapsed_local = time.perf_counter() - t0
    elapsed = elapsed_local
    rank_ms = [1e3 * elapsed_local / a.steps]
    if world > 1:
        cdev = dev if a.backend == "nccl" else "cpu"
        tt = torch.tensor([elapsed_local], dtype=torch.float64, device=cdev)
        allt = [torch.empty_like(tt) for _ in range(world)]
        dist.all_gather(allt, tt)
        rank_ms = [1e3 * float(t.item()) / a.steps for t in allt]
        elapsed = max(float(t.item()) for t in allt)
    kernel_ms = float(np.mean([s.elapsed_time(e) for s, e in ev]))
    solve_form = solver.ctx.last_solve_form()

    status = o["status"].cpu().numpy(); iters = o["iters"].cpu().numpy()
    solved = float(np.mean((status == 1) | (status == 2)))
    it1 = iters + 1                                         # iters holds the index of the last iteration
    mean_iters = float(np.mean(it1))

    def tool(name, argv):
        """One of tools/*.py, in-process (this process prints ONE line)."""
        import contextlib, importlib.util, io
        spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
        mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
        with contextlib.redirect_stdout(io.StringIO()):
            return mod.main(argv)

    # N > 1: BASELINE config 5 on the N GPUs in the same run -- the fleet's agents sharded over the ranks, no collective
    # on the step (tools/mpc_bench.py --gpus N; every rank takes part, rank 0 holds the fleet's figures)
    mpc_multi = None
    if world > 1 and not a.no_secondary:
        try:
            mpc_multi = tool("mpc_bench", ["--steps", "150", "--gpus", str(world), "--backend", a.backend] +
                             (["--share-device"] if a.share_device else []))
        except Exception as e:
            mpc_multi = {"error": repr(e)[:200]}

    out = None
    if rank == 0:
        value = total_candidates * a.steps / elapsed
        alg_bytes = batch.algorithmic_bytes() * B
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        flops = 2.0 * B * S * mean_iters * FLOPS_PER_SEGMENT_ITER
        hbm, hbm_src = measured_profile("hbm", B, S, a.variant, a.workload)
        sq, sq_src = measured_profile("sq", B, S, a.variant, a.workload)
        sq_flops = sq["derived"].get("fp64_flops_per_launch_all_lanes") if sq else None
        hist = np.bincount(it1[(status == 1) | (status == 2)].astype(np.int64), minlength=1)
        out = {
            "metric": "trajectory QP solves/sec (20-seg order-5 corridor)", "value": value, "unit": "solves/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload_label(a.workload, a.batch, S, a.variant, world, a.scaling),
                       "generator": a.workload, "batch_per_gpu": B if a.scaling == "weak" else None,
                       "batch_total": total_candidates, "segments": S, "variant": a.variant,
                       "parallelism": "shard%d" % world,
                       "collective": "none" if world == 1 else "%s all_gather of (cost, index, the local winner's control points) = %d B per rank and step"
                                     % ("rccl" if a.backend == "nccl" else a.backend, 16 + 96 * S)},
            "ms_per_step_by_rank": rank_ms,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         # HBM bytes per launch from the PMC passes (FETCH_SIZE doubled as the guide prescribes for wide
                         # coalesced reads + WRITE_SIZE) of THIS kernel build on THIS workload, else null
                         "traffic": hbm["bytes_per_launch"] if hbm else None, "traffic_source": hbm_src,
                         "kernel": SOLVE_FORMS.get(solve_form, "btrapz::ipm_solve_kernel"), "kernel_ms": kernel_ms, "kernel_source_hash": kernel_stamp(),
                         "algorithmic_bytes_per_solve": batch.algorithmic_bytes(),
                         "workspace_bytes": solver.ctx.workspace_bytes(),
                         "note": "on-chip solve: the binding resource is FP64 VALU issue + dependent sweeps, not HBM "
                                 "(SURVEY 8d); see fp64_valu" + ("; traffic above the algorithmic bytes is the iterate of the candidates the "
                                 "first launch hands to the second (69-74 doubles per segment, written once and read once, for the ~9 % of "
                                 "the axis problems that are handed over) and their records read a second time: not re-reads of a "
                                 "working set" if solve_form in (3, 11) else "") +
                                 ("; the two-wavefronts-per-SIMD form also spills ~13 doubles of read-only problem data per lane to "
                                  "scratch (written once per solve at set-up, ~0.35 GB per launch pair, re-read from L2): DESIGN 3.3" if solve_form in (8, 11) else ""),
                         "fp64_valu": {"achieved_tflops": flops / (kernel_ms * 1e-3) / 1e12, "peak_tflops": FP64_VALU_PEAK_TFLOPS,
                                       "frac": flops / (kernel_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                                       "model": "%.0f useful flops per segment per iteration x %.2f mean iterations" %
                                                (FLOPS_PER_SEGMENT_ITER, mean_iters),
                                       # what the SIMDs actually executed (SQ_INSTS_VALU_{FMA,MUL,ADD,TRANS}_F64 x 64
                                       # lanes, FMA = 2), from the counter profile of this kernel build and workload
                                       "executed_tflops": None if sq_flops is None else sq_flops / (kernel_ms * 1e-3) / 1e12,
                                       "executed_frac": None if sq_flops is None else sq_flops / (kernel_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                                       "valu_busy_frac_of_wave_cycles": sq["derived"].get("frac_of_wave_cycles:SQ_ACTIVE_INST_VALU") if sq else None,
                                       "counter_source": sq_src}},
            "solved_fraction": solved, "mean_ipm_iterations": mean_iters,
            "ipm_iteration_histogram": {str(i): int(c) for i, c in enumerate(hist) if c},
            "status_counts": {str(int(k)): int((status == k).sum()) for k in np.unique(status)},
            "winner": {"index": int(win_idx.item()), "cost": float(win_cost.item()),
                       # every rank holds the winner's control points after the step; their checksum and first values
                       "ctrl_sum": float(wctrl[0].sum().item()), "ctrl_head": [float(v) for v in wctrl[0, :4].tolist()]},
        }
        # The other BASELINE configurations and the other workload family, in the same (driver-timed) run: a short timed
        # loop each, with the figures of the headline -- kernel time of the whole solve call by HIP events, roofline of
        # the algorithmic bytes, useful FP64 rate -- and, where not every candidate has a solution (scenario_1's late slow
        # obstacle; the cuboid variant's empty inscribed intervals, which end before the first iteration), the rate of
        # the SOLVED candidates beside the rate of all.
        def timed_config(label, generator, b2, sh2, variant2):
            d2 = solver.upload(b2)
            for _ in range(2):
                solver.solve(d2, sh2, lean=a.lean)
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = max(3, min(10, a.steps))
            e0.record()
            for _ in range(reps):
                o2 = solver.solve(d2, sh2, lean=a.lean)
            e1.record()
            torch.cuda.synchronize(dev)
            ms2 = e0.elapsed_time(e1) / reps
            form2 = solver.ctx.last_solve_form()
            st2 = o2["status"].cpu().numpy()
            ok2 = (st2 == 1) | (st2 == 2)
            it2 = float(np.mean(o2["iters"].cpu().numpy() + 1))
            B2, S2 = b2.B, b2.S
            gbs = b2.algorithmic_bytes() * B2 / (ms2 * 1e-3) / 1e9
            tfl = 2.0 * B2 * S2 * it2 * FLOPS_PER_SEGMENT_ITER / (ms2 * 1e-3) / 1e12
            return {"workload": label, "generator": generator, "batch": B2, "segments": S2, "variant": variant2,
                    "kernel": SOLVE_FORMS.get(form2, "btrapz::ipm_solve_kernel").split(" (")[0], "kernel_ms": ms2,
                    "solves_per_s_kernel_only": B2 / (ms2 * 1e-3), "solved_fraction": float(ok2.mean()),
                    "solved_solves_per_s_kernel_only": float(ok2.sum()) / (ms2 * 1e-3), "mean_ipm_iterations": it2,
                    "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                 "algorithmic_bytes_per_solve": b2.algorithmic_bytes(),
                                 "fp64_valu": {"achieved_tflops": tfl, "peak_tflops": FP64_VALU_PEAK_TFLOPS, "frac": tfl / FP64_VALU_PEAK_TFLOPS}}}
        out["solved_solves_per_s"] = value * solved
        if world == 1 and not a.no_secondary:
            from spectral_amd import synth as _sy
            other = "generic" if a.workload == "scenario1" else "scenario1"
            b2, sh2 = make_workload(other, B, S, a.variant, 0)
            out["secondary"] = timed_config(workload_label(other, B, S, a.variant, 1, "weak"), other, b2, sh2, a.variant)
            if not (B == 4096 and S == 10 and a.variant == 0):
                b2, sh2 = _sy.make_batch(4096, 10, config=2)
                out["config2"] = timed_config("BASELINE config 2: " + workload_label("generic", 4096, 10, 0, 1, "weak"), "generic", b2, sh2, 0)
            if not (a.variant == 1 and a.workload == "scenario1"):
                b2, sh2 = make_workload("scenario1", B, S, 1, 0)
                out["config4"] = timed_config("BASELINE config 4: " + workload_label("scenario1", B, S, 1, 1, "weak"), "scenario1", b2, sh2, 1)
            del b2
            # Informational, NOT `value`: two INDEPENDENT batches in flight (a serving loop that does not wait for batch i's
            # winner before it launches batch i + 1): two contexts on two streams, the same step each.  What it hides is the
            # tail of a step -- the last round of wavefronts and the resume launch run at a fraction of the machine's width.
            try:
                solver_b = BatchSolver(local_rank)
                streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
                pair = [(solver, db), (solver_b, solver_b.upload(batch))]

                def step_on(k):
                    sv, dbk = pair[k]
                    with torch.cuda.stream(streams[k]):
                        ok = sv.solve(dbk, shared, lean=a.lean)
                        sv.argmin(ok["cost"], index_base=index_base)
                for k in (0, 1, 0, 1):
                    step_on(k)
                torch.cuda.synchronize(dev)
                n2 = 2 * max(a.steps // 2, 4)
                t2 = time.perf_counter()
                for i in range(n2):
                    step_on(i & 1)
                torch.cuda.synchronize(dev)
                dt2 = time.perf_counter() - t2
                out["two_batches_in_flight"] = {"solves_per_s": B * n2 / dt2, "ms_per_batch": 1e3 * dt2 / n2, "batches": n2,
                                                "note": "two independent batches on two streams / contexts; informational -- `value` is one batch at a time"}
                del solver_b, pair
            except Exception as e:
                out["two_batches_in_flight"] = {"error": repr(e)[:200]}
        # the other named configurations, in the same (driver-timed) run: BASELINE config 5 (receding horizon, one GPU)
        # and the knot-level pipeline through the tools that profiles/ documents
        keep = lambda d, keys: {k: d[k] for k in keys if k in d}
        if mpc_multi is not None:
            out["config5_%d_gpus" % world] = keep(mpc_multi, ("workload", "n_gpus", "agents_per_gpu", "parallelism", "achieved_hz",
                                                              "achieved_hz_by_rank", "target_hz", "p50_step_ms", "p99_step_ms",
                                                              "mean_ipm_iterations", "solved_fraction_mean", "candidates_per_s", "error"))
        if world == 1 and not a.no_secondary:
            try:
                m = tool("mpc_bench", ["--steps", "150"])
                out["config5_one_gpu"] = keep(m, ("workload", "achieved_hz", "target_hz", "p50_step_ms", "p99_step_ms",
                                                  "mean_ipm_iterations", "solved_fraction_mean", "candidates_per_s"))
                pl = tool("pipeline_bench", ["--reps", "3"])
                out["pipeline_knots_to_control_points"] = keep(pl, ("workload", "corridor_ms", "ragged_solve_ms", "ragged_solve_form",
                                                                    "end_to_end_candidates_per_s", "solved_fraction", "corridor_roofline"))
                if "solved_fraction" in pl and "end_to_end_candidates_per_s" in pl:
                    out["pipeline_knots_to_control_points"]["end_to_end_solved_candidates_per_s"] = pl["end_to_end_candidates_per_s"] * pl["solved_fraction"]
                pp = tool("pipeline_bench", ["--reps", "3", "--prisms"])
                out["pipeline_prisms_to_control_points"] = keep(pp, ("workload", "prism_ms", "corridor_ms", "ragged_solve_ms", "ragged_solve_form",
                                                                     "end_to_end_scenes_per_s", "solved_fraction", "prism_roofline", "corridor_roofline"))
                if "solved_fraction" in pp and "end_to_end_scenes_per_s" in pp:
                    out["pipeline_prisms_to_control_points"]["end_to_end_solved_scenes_per_s"] = pp["end_to_end_scenes_per_s"] * pp["solved_fraction"]
            except Exception as e:                                  # the headline must not depend on the extras
                out["extras_error"] = repr(e)[:200]
        # p50 latency of ONE solve (B = 1), inputs resident, including the sync
        if a.latency_reps > 0:
            one = solver.upload(batch.slice(0, 1))
            lat = []
            solve_one, _o1 = solver.prepare(one, shared)   # argument structs built once: a call is the C entry point's time
            for i in range(a.latency_reps + 20):
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                solve_one()
                torch.cuda.synchronize(dev)
                lat.append(time.perf_counter() - t1)
            lat = np.array(lat[20:]) * 1e3
            out["p50_solve_latency_ms"] = float(np.percentile(lat, 50))
            out["p99_solve_latency_ms"] = float(np.percentile(lat, 99))
            out["solve_latency_reps"] = int(len(lat))
            # (one candidate runs the split form of the kernel -- one candidate per wavefront, rows over three lanes,
            #  btrapz_options.split -- when it has at most 21 segments; the packed form beside it)
            lat_p = []
            solve_packed, _o2 = solver.prepare(one, shared, split=-1)
            for i in range(a.latency_reps // 2 + 20):
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter(); solve_packed(); torch.cuda.synchronize(dev); lat_p.append(time.perf_counter() - t1)
            out["p50_solve_latency_packed_form_ms"] = float(np.percentile(np.array(lat_p[20:]) * 1e3, 50))
            lat_h = []
            b1 = batch.slice(0, 1)
            for i in range(max(20, a.latency_reps // 4)):
                t1 = time.perf_counter(); solver.ctx.solve_host(b1, shared); lat_h.append(time.perf_counter() - t1)
            out["p50_solve_latency_with_pcie_ms"] = float(np.percentile(np.array(lat_h[5:]) * 1e3, 50))
            # the reference's own call pattern: ONE find_traj per replanning step (cart_frenet.py:1567) -- knot-level
            # input of the same scene (N = 201, two lane corridors) through the host corridor stage and the one-launch
            # single-candidate kernel, trajectory and control points back in host memory (btrapz_find_traj_mem)
            from spectral_amd import synth as _synth
            kb1 = _synth.scenario1_knots(1, S)
            prm = native.CParams(*[float(v) for v in _synth.REFERENCE_WEIGHTS], 0)
            lat_f = []
            call1 = native.TrajCall(a.variant, prm, kb1)   # input struct and output buffers built once: a call is the library's time
            for i in range(max(30, a.latency_reps // 2)):
                t1 = time.perf_counter(); cst, _, ctl = call1(); lat_f.append(time.perf_counter() - t1)
            out["p50_find_traj_mem_ms"] = float(np.percentile(np.array(lat_f[5:]) * 1e3, 50))
            out["find_traj_mem_segments"] = None if ctl is None else int(len(ctl) // 12)
            # ... and on one of the reference's own bundled corridor files (src/c_road_s1_3.txt: N = 71 knots, 3 obstacles,
            # 8 segments; tests/golden/inputs holds the reference's data files) with its weights.txt
            try:
                from spectral_amd import knots as _kn
                gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "inputs")
                kbr = _kn.parse_corridor_file(os.path.join(gdir, "c_road_s1_3.txt"))
                prr = native.CParams(*[float(v) for v in np.loadtxt(os.path.join(gdir, "weights.txt"))], 0)
                callr = native.TrajCall(0, prr, kbr)
                lat_r = []
                for i in range(max(30, a.latency_reps // 2)):
                    t1 = time.perf_counter(); cr, _, ctr = callr(); lat_r.append(time.perf_counter() - t1)
                out["p50_find_traj_mem_c_road_s1_3_ms"] = float(np.percentile(np.array(lat_r[5:]) * 1e3, 50))
                out["find_traj_mem_c_road_s1_3"] = {"segments": None if ctr is None else int(len(ctr) // 12), "cost": float(cr),
                                                    "iterations": int(native.lib().btrapz_find_traj_last_iterations())}
            except OSError:
                pass
            # ... and exactly as the reference's harness calls it: corridor text file in, trajectory text file out
            # (trp_wrapper.py:45-66 -> find_traj(Params*)); the files live in a temporary directory
            import ctypes as _C, tempfile as _tf
            with _tf.TemporaryDirectory() as tdir:
                fin, fout = os.path.join(tdir, "corridor.txt"), os.path.join(tdir, "traj.txt")
                _knots_mod = __import__("spectral_amd.knots", fromlist=["write_corridor_file"])
                _knots_mod.write_corridor_file(fin, kb1)
                lat_t = []
                for i in range(max(30, a.latency_reps // 2)):
                    t1 = time.perf_counter(); native.lib().btrapz_find_traj(a.variant, fin.encode(), fout.encode(), _C.byref(prm)); lat_t.append(time.perf_counter() - t1)
                out["p50_find_traj_file_ms"] = float(np.percentile(np.array(lat_t[5:]) * 1e3, 50))
            # ... and the replanning loop it sits in: consecutive calls on nearly the same scene, each starting from
            # the state the previous call of this thread left on the device (BTRAPZ_WARM=1; opt-in
            # because the last digits of a result then depend on the call history)
            from spectral_amd import knots as _knots
            near = _knots.jittered(kb1, 4, seed=5, s_shift=0.0, l_shift=0.0)        # four copies of the scene ...
            rs = np.random.default_rng(5)                                          # ... whose reference lines differ by centimetres
            near.s_ref += rs.uniform(-0.02, 0.02, (4, 1)) * np.linspace(0, 1, near.N)[None, :]
            near.l_ref += rs.uniform(-0.01, 0.01, (4, 1))
            os.environ["BTRAPZ_WARM"] = "1"
            lat_w, ok_w = [], 0
            calls4 = [native.TrajCall(a.variant, prm, near, j) for j in range(4)]
            for i in range(max(30, a.latency_reps // 2)):
                t1 = time.perf_counter(); cst, _, _ = calls4[i % 4](); lat_w.append(time.perf_counter() - t1)
                ok_w += cst < 1e10
            del os.environ["BTRAPZ_WARM"]
            out["p50_find_traj_mem_replanning_ms"] = float(np.percentile(np.array(lat_w[5:]) * 1e3, 50))
            out["find_traj_mem_replanning_solved"] = ok_w / len(lat_w)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(batch, shared, a.cpu_seconds)
        else:
            out["cpu_baseline"] = None
    if rank == 0:
        line_out.write(json.dumps(out) + "\n"); line_out.flush()     # the line is out before anything below can go wrong
    if world > 1:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception as e:                 # (a rank that lost step with the others must not take the line with it)
            sys.stderr.write("bench.py: rank %d: %r at shutdown\n" % (rank, e))


if __name__ == "__main__":
    main()
