"""profiles/scaling_model_latest.json from an N = 1 bench line (its `scaling_model` block): python tools/save_scaling_model.py bench.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
line = [l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]
sm = json.loads(line)["scaling_model"]
json.dump(sm, open(os.path.join(ROOT, "profiles", "scaling_model_latest.json"), "w"), indent=1, sort_keys=True)
print({k: v.get("predicted_speedup_8_at_30us") for k, v in sm.items() if isinstance(v, dict) and "predicted_speedup_8_at_30us" in v})
