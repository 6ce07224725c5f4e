#!/usr/bin/env python3
"""Counters of the corridor kernel (gpurun_out/<tag>/corridor_pmc{1,2,3}, written by tools/collect_profiles.sh) ->
profiles/<tag>_corridor_pmc.json: per wavefront (= per candidate) averages over the first-pass launches."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    per, kernel = {}, None
    for i in (1, 2, 3):
        files = glob.glob(os.path.join(ROOT, "gpurun_out", tag, "corridor_pmc%d" % i, "*", "*counter_collection.csv"))
        if not files:
            sys.exit("no counter file for pass %d" % i)
        acc, disp = collections.defaultdict(float), set()
        for r in csv.DictReader(open(files[0])):
            if "corridor_" in r["Kernel_Name"] and int(r["Grid_Size"]) >= 65536 * 64:
                kernel = r["Kernel_Name"].split("(")[0]
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
        for c, v in acc.items():
            per[c] = round(v / len(disp) / 65536, 1)
    life = per["SQ_WAVE_CYCLES"]
    out = {"command": "rocprofv3 --pmc <group> -- python3 tools/pipeline_bench.py --reps 2 (three separate passes, counters only)",
           "workload": "65 536 jittered copies of c_road_s1_3.txt (N = 71 knots, 3 obstacles), first-pass launches of " + str(kernel),
           "per_wavefront": per,
           "derived": {"wave_lifetime_quad_cycles": life,
                       "valu_active_frac_of_lifetime": round(per["SQ_ACTIVE_INST_VALU"] / life, 3),
                       "scalar_active_frac_of_lifetime": round(per["SQ_ACTIVE_INST_SCA"] / life, 3),
                       "wait_any_frac": round(per["SQ_WAIT_ANY"] / life, 3),
                       "fp64_instructions": per["SQ_INSTS_VALU_ADD_F64"] + per["SQ_INSTS_VALU_MUL_F64"] +
                                            per["SQ_INSTS_VALU_FMA_F64"] + per["SQ_INSTS_VALU_TRANS_F64"]}}
    path = os.path.join(ROOT, "profiles", "%s_corridor_pmc.json" % tag)
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out["derived"]), "->", path)


if __name__ == "__main__":
    main()
