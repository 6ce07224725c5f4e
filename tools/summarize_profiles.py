#!/usr/bin/env python3
"""Turn gpurun_out/<tag>/ (written by tools/collect_profiles.sh) into the committed files under profiles/.

  python tools/summarize_profiles.py r01

Writes profiles/<tag>_bench.json, <tag>_bench_under_rocprof.json, <tag>_bench_kernel_stats.csv,
<tag>_pmc_hbm.json (FETCH_SIZE / WRITE_SIZE, per launch of the dominant kernel) and <tag>_pmc_sq.json
(SQ counters per launch and per wavefront)."""
import csv
import glob
import json
import os
import shutil
import sys

KERNEL = "ipm_solve_"        # every kernel of a solve call: ipm_solve_kernel, or ipm_solve_capped_kernel + ipm_solve_resume_kernel
PRIMARY = ("ipm_solve_kernel", "ipm_solve_capped_kernel", "ipm_solve_queue_kernel", "ipm_solve_split_kernel",
           "ipm_solve_lean_kernel", "ipm_solve_lean_capped_kernel")   # one launch of these = one solve


def counter_means(root, pattern="pmc_*", kernel=KERNEL):
    """{counter: (mean over launches of the kernel, launches, resource columns)} over every matching directory."""
    out = {}
    newest = {}   # gpurun's merge keeps earlier runs' files: take the newest CSV of every counter group
    for path in glob.glob(os.path.join(root, pattern, "**", "*counter_collection.csv"), recursive=True):
        group = os.path.relpath(path, root).split(os.sep)[0]
        if group not in newest or os.path.getmtime(path) > os.path.getmtime(newest[group]):
            newest[group] = path
    for path in newest.values():
        acc = {}
        solve_of = {}      # dispatch -> number of the solve call it belongs to (a primary kernel opens one)
        nsolve = 0
        rows = list(csv.DictReader(open(path, newline="")))
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        for row in rows:
                if kernel not in row["Kernel_Name"]:
                    continue
                base = row["Kernel_Name"].split("(")[0].split("::")[-1]
                primary = kernel != KERNEL or base in PRIMARY
                if row["Dispatch_Id"] not in solve_of:
                    if primary:
                        nsolve += 1
                    solve_of[row["Dispatch_Id"]] = max(nsolve, 1)
                key = solve_of[row["Dispatch_Id"]]
                acc.setdefault(row["Counter_Name"], {}).setdefault(key, 0.0)
                acc[row["Counter_Name"]][key] += float(row["Counter_Value"])
                # rocprofv3's register columns are NOT the compiler's VGPR / AGPR counts: on gfx950 it reports the unified
                # allocation (VGPR + AGPR, rounded up to the granule of 8) divided by two under VGPR_Count and 0 under
                # Accum_VGPR_Count (256 + 199 -> 456 / 2 = 228).  Kept under names that say so; the compiler's figures
                # are added by compiler_resources() below.
                if primary:
                    res = dict(rocprof_VGPR_Count_field=row["VGPR_Count"], rocprof_Accum_VGPR_Count_field=row["Accum_VGPR_Count"],
                               sgpr=row["SGPR_Count"], lds=row["LDS_Block_Size"], scratch=row["Scratch_Size"], grid=row["Grid_Size"],
                               primary_kernel=base)
        for name, per in acc.items():
            vals = list(per.values())
            out[name] = dict(mean=sum(vals) / len(vals), min=min(vals), max=max(vals), launches=len(vals), **res)
    return out


def compiler_resources(here, kernel="ipm_solve_kernelE"):
    """VGPR / AGPR / scratch / LDS of the kernel as the compiler reports them (tools/kernel_resources.sh)."""
    import re
    import subprocess
    try:
        txt = subprocess.run(["bash", os.path.join(here, "tools", "kernel_resources.sh")], capture_output=True, text=True, timeout=900).stdout
    except Exception:
        return None
    for line in txt.splitlines():
        if kernel in line:
            g = lambda pat: int(re.search(pat + r"\s*(\d+)", line).group(1))
            return dict(vgpr=g(r"VGPRs:"), agpr=g(r"AGPRs:"), sgpr=g(r"TotalSGPRs:"), scratch_bytes_per_lane=g(r"ScratchSize \[bytes/lane\]:"),
                        lds_bytes_per_block=g(r"LDS Size \[bytes/block\]:"), waves_per_simd=g(r"Occupancy \[waves/SIMD\]:"))
    return None


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src, dst = os.path.join(here, "gpurun_out", tag), os.path.join(here, "profiles")
    failed = os.path.join(src, "failed.txt")
    if os.path.exists(failed) and os.path.getsize(failed) and "--allow-failed" not in sys.argv:
        sys.exit("collect_profiles.sh recorded failures (pass --allow-failed to summarise what is there):\n" + open(failed).read())
    for need in ("bench.json", "bench_under_rocprof.json", "kernel_source_hash.txt"):
        if not os.path.exists(os.path.join(src, need)) or not os.path.getsize(os.path.join(src, need)):
            sys.exit("missing or empty %s under %s" % (need, src))
    stamp = open(os.path.join(src, "kernel_source_hash.txt")).read().strip()
    shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, f"{tag}_bench.json"))
    shutil.copy(os.path.join(src, "bench_under_rocprof.json"), os.path.join(dst, f"{tag}_bench_under_rocprof.json"))
    stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(max(stats, key=os.path.getmtime), os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))
    for sub, name in (("trace_pipeline", "pipeline"), ("trace_pipeline_s1", "pipeline_scenario1"), ("trace_pipeline_prisms", "pipeline_prisms"),
                      ("trace_find_traj", "find_traj"), ("trace_find_traj_packed", "find_traj_packed")):
        pstats = glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True)
        if pstats:
            shutil.copy(max(pstats, key=os.path.getmtime), os.path.join(dst, f"{tag}_{name}_kernel_stats.csv"))
    for name in ("mpc_warm", "mpc_cold", "mpc_warm_minfirst05", "mpc_cold_minfirst05", "pipeline", "pipeline_scenario1", "pipeline_prisms",
                 "pipeline_cap8", "pipeline_scenario1_cap8", "pipeline_prisms_cap8", "cap_bench",
                 "bench_config2", "bench_config4", "bench_generic", "bench_2rank_gloo_strong", "split_bench", "mpc_warm_2rank_gloo",
                 "lean_bench", "lean_bench_B512", "lean_bench_B2048", "lean_bench_B8192", "lean_bench_B16384", "cap_bench_lean", "sibling_bench"):
        if os.path.exists(os.path.join(src, name + ".json")):
            # the tools print ONE JSON line; libraries may print before it (gloo announces its ranks on stdout)
            lines = [l for l in open(os.path.join(src, name + ".json")).read().splitlines() if l.lstrip().startswith("{")]
            if not lines:
                sys.exit("no JSON line in %s/%s.json" % (src, name))
            with open(os.path.join(dst, f"{tag}_{name}.json"), "w") as fh:
                fh.write(lines[-1] + "\n")
    bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
    wl = dict(bench["config"]); wl["workload"] = wl.get("generator", "generic")      # the key bench.py matches on
    wl["label"] = bench["config"]["workload"]
    if wl.get("batch_per_gpu") is None:
        wl["batch_per_gpu"] = wl.get("batch_total")
    c = counter_means(src)
    if not c:
        sys.exit("no counter CSVs under %s/pmc_*" % src)

    # calibration of FETCH_SIZE / WRITE_SIZE in the solve kernel's access widths (tools/fetch_calib.hip)
    calib = {}
    for mode, counter, kern in (("fetch_8", "FETCH_SIZE", "read8"), ("fetch_16", "FETCH_SIZE", "read16"), ("write_48", "WRITE_SIZE", "write48")):
        cc = counter_means(src, "calib_" + mode, kern)
        js = os.path.join(src, "calib_%s.json" % mode)
        if counter in cc and os.path.exists(js):
            known = json.loads([l for l in open(js).read().splitlines() if l.startswith("{")][-1])["bytes_per_launch"]
            calib[mode] = dict(known_bytes_per_launch=known, counter_KB=cc[counter]["mean"], launches=cc[counter]["launches"],
                               bytes_per_counted_byte=known / (cc[counter]["mean"] * 1024.0))

    hbm = {}
    for k in ("FETCH_SIZE", "WRITE_SIZE"):
        if k in c:
            hbm[k] = dict(launches=c[k]["launches"], mean_KB=c[k]["mean"], min_KB=c[k]["min"], max_KB=c[k]["max"],
                          **{r: c[k][r] for r in ("rocprof_VGPR_Count_field", "rocprof_Accum_VGPR_Count_field", "sgpr", "lds", "scratch", "grid", "primary_kernel")})
    primary = next(iter(c.values())).get("primary_kernel", "ipm_solve_kernel")
    cres = compiler_resources(here, primary + "E")
    if len(hbm) == 2:
        hbm["compiler_resources"] = cres
        hbm["workload"] = wl
        hbm["kernel_source_hash"] = stamp
        hbm["calibration"] = calib
        hbm["bytes_per_launch_raw"] = (hbm["FETCH_SIZE"]["mean_KB"] + hbm["WRITE_SIZE"]["mean_KB"]) * 1024.0
        # corrected as MI355X_MICROARCH.md (HBM) prescribes: counters times the factor measured on a known byte count in
        # the kernel's own access width (8 B per lane reads; 8 B stores at a 48 B lane stride); without a calibration
        # run the guide's factor for wide coalesced reads (2) and 1 for the stores
        kf = calib.get("fetch_8", {}).get("bytes_per_counted_byte", 2.0)
        kw = calib.get("write_48", {}).get("bytes_per_counted_byte", 1.0)
        hbm["fetch_factor"], hbm["write_factor"] = kf, kw
        hbm["bytes_per_launch"] = (kf * hbm["FETCH_SIZE"]["mean_KB"] + kw * hbm["WRITE_SIZE"]["mean_KB"]) * 1024.0
        hbm["note"] = ("FETCH_SIZE/WRITE_SIZE in KB, separate --pmc passes (TCC slot limit), means over the launches of "
                       "the solve call's kernels (btrapz::ipm_solve_*: one launch, or the capped launch + the resume launch).  bytes_per_launch = fetch_factor x FETCH_SIZE + write_factor x WRITE_SIZE with the "
                       "factors measured by tools/fetch_calib.hip on known byte counts in the kernel's access widths "
                       "(calibration); bytes_per_launch_raw is the uncorrected sum.")
        json.dump(hbm, open(os.path.join(dst, f"{tag}_pmc_hbm.json"), "w"), indent=1)

    sq = {k: v["mean"] for k, v in c.items() if k.startswith("SQ_") or k.startswith("GRBM")}
    if "SQ_WAVES" in sq:
        w = sq["SQ_WAVES"]
        per_wave = {k: v / w for k, v in sq.items() if k.startswith("SQ_INSTS")}
        derived = dict(per_wave_instructions=per_wave)
        if "SQ_WAVE_CYCLES" in sq:
            wc = sq["SQ_WAVE_CYCLES"]  # in units of 4 cycles, like the SQ_ACTIVE_* / SQ_WAIT_* counters
            derived["wave_lifetime_cycles"] = 4.0 * wc / w
            for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA",
                      "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS"):
                if k in sq:
                    derived["frac_of_wave_cycles:" + k] = sq[k] / wc
            if "SQ_ACTIVE_INST_VALU" in sq and "SQ_INSTS_VALU" in sq:
                derived["cycles_per_valu_instruction"] = 4.0 * sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_INSTS_VALU"]
        f64 = {k: sq[k] for k in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64",
                                  "SQ_INSTS_VALU_TRANS_F64") if k in sq}
        if f64:
            # wave-level instruction counts; one instruction = 64 lanes; FMA = 2 flops
            flops = 64.0 * (2 * f64.get("SQ_INSTS_VALU_FMA_F64", 0) + f64.get("SQ_INSTS_VALU_MUL_F64", 0) +
                            f64.get("SQ_INSTS_VALU_ADD_F64", 0) + f64.get("SQ_INSTS_VALU_TRANS_F64", 0))
            derived["fp64_flops_per_launch_all_lanes"] = flops
            derived["fp64_share_of_valu_instructions"] = sum(f64.values()) / sq["SQ_INSTS_VALU"]
        json.dump(dict(kernel=primary + (" + ipm_solve_resume_kernel" if primary == "ipm_solve_capped_kernel" else " + ipm_solve_lean_resume_kernel" if primary == "ipm_solve_lean_capped_kernel" else ""), workload=wl, kernel_source_hash=stamp, compiler_resources=cres, per_launch=sq, derived=derived,
                       note="rocprofv3 --pmc passes (tools/collect_profiles.sh), means over the launches of one "
                            "bench.py --steps 3 --warmup 1 run; SQ cycle counters are in units of 4 clock cycles."),
                  open(os.path.join(dst, f"{tag}_pmc_sq.json"), "w"), indent=1)
    # L2 behaviour per solve kernel (round 5; the optional counter groups of collect_profiles.sh): mean per launch
    l2 = {}
    for kern in ("ipm_solve_lean_capped_kernel", "ipm_solve_lean_resume_kernel"):
        per = {}
        for path in glob.glob(os.path.join(src, "pmc_opt_*", "**", "*counter_collection.csv"), recursive=True):
            sums, launches = {}, {}
            for row in csv.DictReader(open(path, newline="")):
                if kern + "(" not in row["Kernel_Name"] and not row["Kernel_Name"].split("(")[0].endswith(kern):
                    continue
                sums[row["Counter_Name"]] = sums.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                launches.setdefault(row["Counter_Name"], set()).add(row["Dispatch_Id"])
            for k, v in sums.items():
                # (gpurun's merge keeps earlier runs' files: the newest CSV wins)
                if k not in per or os.path.getmtime(path) > per[k][1]:
                    per[k] = (v / max(len(launches[k]), 1), os.path.getmtime(path))
        if per:
            d = {k: v[0] for k, v in sorted(per.items())}
            if d.get("TCC_REQ_sum"):
                d["l2_hit_fraction"] = d.get("TCC_HIT_sum", 0.0) / d["TCC_REQ_sum"]
            if d.get("TCP_TCC_READ_REQ_sum"):
                d["read_requests_reaching_memory_fraction"] = d.get("TCC_EA0_RDREQ_sum", 0.0) / d["TCP_TCC_READ_REQ_sum"]
            if d.get("TCP_TCC_WRITE_REQ_sum"):
                d["write_requests_reaching_memory_fraction"] = d.get("TCC_EA0_WRREQ_sum", 0.0) / d["TCP_TCC_WRITE_REQ_sum"]
            l2[kern] = d
    if l2:
        json.dump(dict(note="L2 (TCC) and L1->L2 (TCP_TCC) request counters of the two solve kernels of BASELINE config 3 (65 536 x 20, lean "
                            "form, two launches), rocprofv3 --pmc in separate passes (tools/collect_profiles.sh quick), mean per launch.  "
                            "TCC_EA0_* are the L2's memory-side requests (what FETCH_SIZE / WRITE_SIZE derive from).",
                       kernels=l2, kernel_source_hash=stamp), open(os.path.join(dst, f"{tag}_pmc_l2.json"), "w"), indent=1)
    print("wrote profiles for", tag, "counters:", sorted(c))


if __name__ == "__main__":
    main()
