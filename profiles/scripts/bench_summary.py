import json, sys
for f in sys.argv[1:]:
    d = json.load(open(f)); o = d["other_modes"]
    print(round(d["value"]), d["ms_per_step"], {k: (round(v["pairs_per_s"]) if isinstance(v, dict) and "pairs_per_s" in v else None) for k, v in o.items()})
    print(" streamed:", {k: o["streamed"].get(k) for k in ("chunk_frames", "lanes", "h2d_GBps", "equals_batched_call", "chunk_latency_ms")})
    print(" latency:", o.get("latency"))
