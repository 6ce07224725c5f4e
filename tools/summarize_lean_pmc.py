#!/usr/bin/env python3
"""Counters per launch and per wavefront of every solve kernel in gpurun_out/<tag>/pmc*/ (tools/lean_pmc.sh): one JSON
object {kernel: {counter: mean per launch, ..., per_wave: {...}, derived: {...}}}.

    python tools/summarize_lean_pmc.py <tag> [out.json]
"""
import csv
import glob
import json
import os
import sys


def main():
    tag = sys.argv[1]
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    root = os.path.join(here, "gpurun_out", tag)
    acc = {}
    newest = {}   # gpurun's merge keeps the files of earlier runs: the newest CSV of every pass counts
    for path in glob.glob(os.path.join(root, "pmc*", "**", "*counter_collection.csv"), recursive=True):
        group = os.path.relpath(path, root).split(os.sep)[0]
        if group not in newest or os.path.getmtime(path) > os.path.getmtime(newest[group]):
            newest[group] = path
    for path in newest.values():
        for row in csv.DictReader(open(path, newline="")):
            name = row["Kernel_Name"].split("(")[0].split("::")[-1]
            if not (name.startswith("ipm_") or name.startswith("single_")):
                continue
            d = acc.setdefault(name, {}).setdefault(row["Counter_Name"], {})
            d[row["Dispatch_Id"]] = d.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
            acc[name].setdefault("_res", dict(lds=row["LDS_Block_Size"], scratch=row["Scratch_Size"], grid=row["Grid_Size"],
                                              rocprof_vgpr_field=row["VGPR_Count"], sgpr=row["SGPR_Count"]))
    out = {}
    for name, counters in acc.items():
        rec = {"resources": counters.pop("_res")}
        mean = {c: sum(v.values()) / len(v) for c, v in counters.items()}
        rec["launches"] = {c: len(v) for c, v in counters.items()}.get("SQ_WAVES", 0)
        rec["per_launch"] = mean
        w = mean.get("SQ_WAVES")
        if w:
            rec["per_wave"] = {c: v / w for c, v in mean.items() if c.startswith("SQ_") and c != "SQ_WAVES"}
            pw = rec["per_wave"]
            d = {}
            if "SQ_WAVE_CYCLES" in pw and "SQ_ACTIVE_INST_VALU" in pw:
                # SQ_WAVE_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* count in units of 4 cycles per wave on gfx9: ratios are what matter
                d["valu_busy_share_of_wave_cycles"] = pw["SQ_ACTIVE_INST_VALU"] / pw["SQ_WAVE_CYCLES"]
                d["wait_any_share"] = pw.get("SQ_WAIT_ANY", 0.0) / pw["SQ_WAVE_CYCLES"]
                d["wait_inst_any_share"] = pw.get("SQ_WAIT_INST_ANY", 0.0) / pw["SQ_WAVE_CYCLES"]
            if "SQ_INSTS_VALU" in pw and "SQ_INSTS_VALU_FMA_F64" in pw:
                f64 = pw["SQ_INSTS_VALU_FMA_F64"] + pw["SQ_INSTS_VALU_MUL_F64"] + pw["SQ_INSTS_VALU_ADD_F64"] + pw["SQ_INSTS_VALU_TRANS_F64"]
                d["fp64_share_of_valu_instructions"] = f64 / pw["SQ_INSTS_VALU"]
                d["fp64_flop_per_wave"] = 64 * (2 * pw["SQ_INSTS_VALU_FMA_F64"] + pw["SQ_INSTS_VALU_MUL_F64"] + pw["SQ_INSTS_VALU_ADD_F64"] + pw["SQ_INSTS_VALU_TRANS_F64"])
            rec["derived"] = d
        out[name] = rec
    txt = json.dumps(out, indent=1, sort_keys=True)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
