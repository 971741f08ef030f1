"""Work accounting of the primary tracer on the bench workload (COUNT_WORK context): python tools/primary_stats.py [out.json]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ugrt, bench
s = bench.load_scene(ugrt, 'crash', 1.0, 0)
setup = ugrt.FrameSetup.from_scene(s)
ctx = ugrt.Context(1920, 1080, light_grid=(128, 128), flags=ugrt.FLAG_SHADOW_ALL_CHUNKS | ugrt.FLAG_COUNT_WORK, uniform_dims=(128, 128, 64))
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
r = ugrt.Renderer(ctx, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"])
r.display(setup, reflect=True)
ctx.synchronize()
d = ctx.stats_primary()
d["lanes_per_round"] = round(d["lane_tests"] / max(1, d["rounds"]), 2)
print(json.dumps(d, indent=1))
if len(sys.argv) > 1:
    json.dump(d, open(sys.argv[1], "w"), indent=1)
