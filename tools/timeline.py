import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ugrt
import bench
s = bench.load_scene(ugrt, 'crash', 1.0, 0)
setup = ugrt.FrameSetup.from_scene(s)
ctx = ugrt.Context(1920,1080, light_grid=(128,128), flags=ugrt.FLAG_SHADOW_ALL_CHUNKS, uniform_dims=(128,128,64))
r = ugrt.Renderer(ctx, s["verts"], s["faces"], s["matidx"], s["mat_list"], s["reflect"])
for _ in range(3): r.display(setup, reflect=True)
ctx.synchronize()
# monkeypatch ctx methods to time them (host time incl. sync after)
import types, collections
acc = collections.OrderedDict()
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t0=time.perf_counter(); out=f(*a, **k); t1=time.perf_counter(); ctx.synchronize(); t2=time.perf_counter()
        e=acc.setdefault(name,[0.0,0.0,0]); e[0]+=t1-t0; e[1]+=t2-t1; e[2]+=1
        return out
    setattr(obj, name, g)
for nm in ['set_light_position','upload_camera','grid_build_perspective','grid_arrays','trace_primary','map_rays_to_light','grid_build_spherical','sort_rays','trace_shadow','reflect_rays','grid_build_uniform','trace_dda','shade_reflect','shade_add_shadows']:
    wrap(ctx, nm)
K=10
t0=time.perf_counter()
for _ in range(K): r.display(setup, reflect=True)
ctx.synchronize(); tot=time.perf_counter()-t0
print('frame ms (with syncs after each call):', tot/K*1e3)
for k,(h,d,c) in acc.items(): print('%-26s host %.3f ms  drain %.3f ms  calls/frame %.1f'%(k, h/K*1e3, d/K*1e3, c/K))
print('sum host+drain', sum(h+d for h,d,c in acc.values())/K*1e3)
