import json, sys
for f in sys.argv[1:]:
    lines = [l for l in open(f).read().strip().splitlines() if l.startswith('{"metric"')]
    if not lines:
        print(f, "NO JSON"); print(open(f).read()[-1500:]); continue
    d = json.loads(lines[-1])
    print(f, "value=%.0f edges/s  ms/step=%.3f  n_gpus=%d" % (d["value"], d["ms_per_step"], d["n_gpus"]))
    print("  kernels:", {k: round(v["avg_us"], 1) for k, v in d["kernels"].items()})
    if d.get("roofline"): print("  roofline:", {k: (round(v, 5) if isinstance(v, float) else v) for k, v in d["roofline"].items() if k != "note"})
    if d.get("cpu_baseline"): print("  cpu:", round(d["cpu_baseline"]["value"]), "edges/s on", d["cpu_baseline"]["cores"], "threads")
