#!/usr/bin/env python3
"""A/B of libbtrapz_hip.so variants on one GPU box: scratch/ab.py name=path ...  (each in its own process)"""
import hashlib, json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def worker():
    import numpy as np, torch
    from spectral_amd import synth
    from spectral_amd.solver import BatchSolver
    reps = int(os.environ.get("AB_REPS", "10")); B = int(os.environ.get("AB_BATCH", "65536"))
    solver = BatchSolver(0); dev = torch.device("cuda:0")
    out = {}
    cases = [("s1x20", lambda: synth.make_scenario1_batch(B, 20, 0)), ("genx20", lambda: synth.make_batch(B, 20, config=3)),
             ("cubx20", lambda: synth.make_scenario1_batch(B, 20, 1)), ("genx10", lambda: synth.make_batch(B, 10, config=2))]
    for ci in [int(x) for x in os.environ.get("AB_CASES", "0,1").split(",")]:
        label, make = cases[ci]
        batch, sh = make(); db = solver.upload(batch)
        eps = float(os.environ.get("AB_EPS", "0"))
        for name, kw in (("one", dict(lean=1, cap_iter=-1, eps=eps)), ("two", dict(lean=1, cap_iter=int(os.environ.get("AB_CAP", "6")), eps=eps))):
            for _ in range(3): o = solver.solve(db, sh, split=-1, **kw)
            torch.cuda.synchronize(dev)
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps): o = solver.solve(db, sh, split=-1, **kw)
                e1.record(); torch.cuda.synchronize(dev)
                best = min(best, e0.elapsed_time(e1) / reps)
            st = o["status"].cpu().numpy(); ok = st > 0
            if os.environ.get("AB_SAVE") and name == "one":
                np.savez(os.path.join(os.environ["AB_SAVE"], f"{label}.npz"), ctrl=o["ctrl"].cpu().numpy(), status=st, iters=o["iters"].cpu().numpy(), cost=o["cost"].cpu().numpy())
            its = o["iters"].cpu().numpy() + 1
            h = hashlib.sha256(o["ctrl"].cpu().numpy()[ok].tobytes() + st.tobytes() + o["iters"].cpu().numpy().tobytes()).hexdigest()[:12]
            out[f"{label}.{name}"] = {"ms": round(best, 4), "hash": h, "solved": int(ok.sum()), "it": float(o["iters"].float().mean()), "itmax": int(its.max()), "it13": int((its >= 13).sum()), "it20": int((its >= 20).sum())}
    print(json.dumps(out))

if __name__ == "__main__":
    if os.environ.get("AB_WORKER"):
        worker(); sys.exit(0)
    res = {}
    for arg in sys.argv[1:]:
        name, path = arg.split("=", 1)
        env = dict(os.environ, AB_WORKER="1")
        if "@" in path:   # name=path@eps: the same library at another tolerance
            path, env["AB_EPS"] = path.split("@", 1)
        if os.environ.get("AB_COMPARE"):
            d = os.path.join("/tmp", "ab_" + name); os.makedirs(d, exist_ok=True); env["AB_SAVE"] = d
        if path != "default": env["BTRAPZ_HIP_LIB"] = os.path.join(ROOT, path)
        p = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True)
        try: res[name] = json.loads(p.stdout.strip().splitlines()[-1])
        except Exception: res[name] = {"error": p.stderr[-800:]}
        print(name, json.dumps(res[name]), flush=True)
    keys = sorted({k for r in res.values() for k in r if k != "error"})
    for k in keys:
        print(k, "  ".join(f"{n}:{r.get(k,{}).get('ms','-')}/{r.get(k,{}).get('hash','-')[:6]}" for n, r in res.items()))

    if os.environ.get("AB_COMPARE") and len(sys.argv) > 2:
        import numpy as np, glob
        names = [a.split("=", 1)[0] for a in sys.argv[1:]]
        ref = names[0]
        for f in sorted(glob.glob(os.path.join("/tmp", "ab_" + ref, "*.npz"))):
            r = np.load(f)
            for n in names[1:]:
                g = os.path.join("/tmp", "ab_" + n, os.path.basename(f))
                if not os.path.exists(g): continue
                v = np.load(g)
                both = (r["status"] > 0) & (v["status"] > 0)
                scale = np.abs(r["ctrl"][both]).max(axis=1)
                dev = np.abs(v["ctrl"][both] - r["ctrl"][both]).max(axis=1) / scale
                print(os.path.basename(f), n, "vs", ref, "accept lost", int(((r["status"] > 0) & ~(v["status"] > 0)).sum()), "gained", int((~(r["status"] > 0) & (v["status"] > 0)).sum()),
                      "status differs", int((r["status"] != v["status"]).sum()), "max rel ctrl dev %.2e" % dev.max(), "p999 %.2e" % np.quantile(dev, 0.999),
                      "mean iters %.3f -> %.3f" % (r["iters"][both].mean() + 1, v["iters"][both].mean() + 1), "argmin same", bool(np.argmin(np.where(r["status"] > 0, r["cost"], np.inf)) == np.argmin(np.where(v["status"] > 0, v["cost"], np.inf))))
