import json,sys
for f in sys.argv[1:]:
    d=json.load(open(f))
    print(f)
    for k,v in d.items():
        print('  %-26s'%k, {n.replace('_launch','').replace('es',''): round(v[n]["solve_ms"],3) for n in ("packed_one_launch","lean_one_launch","packed_two_launches","lean_two_launches")}, 'acc', v["lean_one_launch"]["accept_differs"], 'bit', v["lean_two_launches"]["bit_identical_to_lean_one_launch"], v.get("lean_vs_oracle",{}).get('max_rel_ctrl'))
