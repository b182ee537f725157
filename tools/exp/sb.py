# one-line summaries of bench JSON files (value, legs, kernels)
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads([l for l in open(f) if l.startswith('{"metric"')][-1])
    except Exception as ex:
        print(f, "NO JSON", ex); continue
    def K(x): return {k.replace("tppr_", "").replace("_update", "").replace("store_messages", "msg"): round(v["avg_us"], 1) for k, v in x["kernels"].items()}
    print("%s: %.4f ms/step %.2f M/s  scorer %.4f  host %.3f chain %.3f  %s lat %.2f" % (
        f.split("/")[-1], d["ms_per_step"], d["value"] / 1e6, (d.get("with_scorer") or {}).get("ms_per_step", 0),
        d.get("host_enqueue_ms_per_step") or 0, d.get("chain_bound_ms_per_step") or 0, K(d),
        (d["roofline"].get("latency_model") or {}).get("frac", 0)))
    for w, x in (d.get("workloads") or {}).items():
        print("    %s %.4f ms/step %.2f M/s scorer %.4f %s lat %s" % (w, x["ms_per_step"], x["value"] / 1e6,
              (x.get("with_scorer") or {}).get("ms_per_step", 0), K(x), (x["roofline"].get("latency_model") or {}).get("frac")))
