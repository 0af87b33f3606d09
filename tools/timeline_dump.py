"""One replayed decode step as a per-launch timeline (in-kernel stamps): tools/timeline_dump.py [batch] [ctx] [bf16|fp8] [out.json]"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pygpukit_amd.llm import synthetic as S
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
P = int(sys.argv[2]) if len(sys.argv) > 2 else 150
fmt = sys.argv[3] if len(sys.argv) > 3 else "bf16"
out = sys.argv[4] if len(sys.argv) > 4 else None
cfg = dict(S.QWEN3_0_6B)
w = S.make_qwen3_weights(cfg, seed=0)
eng = S.build_engine_from_weights(cfg, w, max_seq_len=P + 64, max_batch=B, weight_format=fmt)
pr = np.random.default_rng(1).integers(0, cfg["vocab_size"], (B, P))
first = [int(np.argmax(eng.prefill([int(t) for t in pr[b]], seq=b))) for b in range(B)]
eng.set_state(first, [P] * B)
tl = eng.timeline(B, warm=3)
rows = []
for a, b in zip(tl, tl[1:] + [None]):
    rows.append({"kernel": a["kernel"], "workgroups": a["workgroups"], "first_start_us": round(a["first_start_us"], 2), "last_start_us": round(a["last_start_us"], 2),
                 "first_end_us": round(a["first_end_us"], 2), "last_end_us": round(a["last_end_us"], 2), "span_us": round(a["last_end_us"] - a["first_start_us"], 2),
                 "gap_to_next_us": round(b["first_start_us"] - a["last_end_us"], 2) if b else None})
step = rows[-1]["last_end_us"] - rows[0]["first_start_us"]
by = {}
for r in rows:
    d = by.setdefault(r["kernel"], {"launches": 0, "span_us": 0.0, "gap_after_us": 0.0})
    d["launches"] += 1; d["span_us"] += r["span_us"]; d["gap_after_us"] += r["gap_to_next_us"] or 0.0
summary = {k: {"launches": v["launches"], "mean_span_us": round(v["span_us"] / v["launches"], 2), "mean_gap_after_us": round(v["gap_after_us"] / v["launches"], 2),
               "share_of_step": round((v["span_us"] + v["gap_after_us"]) / step, 3)} for k, v in by.items()}
doc = {"what": "one graph-replayed decode step, Qwen3-0.6B %s, batch %d, context %d: per launch the first / last workgroup start and end (s_memrealtime stamps written by every workgroup; pgk_engine_timeline)" % (fmt, B, P),
       "step_us_first_start_to_last_end": round(step, 2), "sum_of_spans_us": round(sum(r["span_us"] for r in rows), 2), "sum_of_gaps_us": round(sum(r["gap_to_next_us"] or 0 for r in rows), 2),
       "by_kernel": summary, "launches": rows}
print(json.dumps({k: doc[k] for k in ("step_us_first_start_to_last_end", "sum_of_spans_us", "sum_of_gaps_us", "by_kernel")}, indent=1))
if out:
    json.dump(doc, open(out, "w"), indent=1)
