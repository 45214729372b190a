"""ESL-like frames: one frame per call vs groups of B frames per call (device-resident AoS), kernel-bound throughput."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.cuda.init()
from x_maps_amd import XMapsEngine, rig

B = int(os.environ.get("B", 16))
G = int(os.environ.get("G", 3))
cp, tables, _, _ = rig.make_esl_like(row_stride=13, device=0)
nf = B * G
host = [rig.render_events(cp, tables, row_stride=13, seed=f)[0] for f in range(nf)]
dev = torch.device("cuda:0")
eng = XMapsEngine(tables, device=0, n_slots=max(4, B * G))
H, W = eng.out_h, eng.out_w
depth = torch.empty((B * G, H, W), dtype=torch.float32, device=dev)
bgr = torch.empty((B * G, H, W, 3), dtype=torch.uint8, device=dev)
aos = [torch.from_numpy(np.concatenate([np.frombuffer(e.tobytes(), np.uint8) if e.dtype.itemsize == 16 else None for e in host[g * B:(g + 1) * B]]).copy()).to(dev) for g in range(G)]
offs = [np.concatenate([[0], np.cumsum([len(e) for e in host[g * B:(g + 1) * B]])]).astype(np.uint64) for g in range(G)]
single = [torch.from_numpy(np.frombuffer(e.tobytes(), np.uint8).copy()).to(dev) for e in host]
torch.cuda.synchronize()
nev = float(np.mean([len(e) for e in host]))

def run_groups(n):
    for i in range(n):
        g = i % G
        eng.process_events_batch_device(aos[g].data_ptr(), offs[g], depth[g * B].data_ptr(), bgr[g * B].data_ptr())
    eng.sync()

def run_single(n):
    for i in range(n):
        f = i % nf
        eng.process_events_device(single[f].data_ptr(), len(host[f]), False, depth[f % 4].data_ptr(), bgr[f % 4].data_ptr())
    eng.sync()

MODES = (("single", run_single, 1), ("groups", run_groups, B))
if os.environ.get("ONLY"):
    MODES = tuple(m for m in MODES if m[0] == os.environ["ONLY"])
for name, fn, per in MODES:
    fn(50)
    c0 = time.perf_counter(); n = 400 if per == 1 else 100; fn(n); dt = time.perf_counter() - c0
    print(f"{name}: {dt / (n * per) * 1e6:.2f} us/frame  {nev * n * per / dt / 1e9:.2f} Gev/s  paths {eng.path_counts()}")
if os.environ.get("ONLY"):
    eng.close(); sys.exit(0)
run_groups(G)
d_g = depth.clone(); b_g = bgr.clone()
for f in range(nf):
    eng.process_events_device(single[f].data_ptr(), len(host[f]), False, depth[f].data_ptr(), bgr[f].data_ptr()); eng.sync()
print("group == single:", bool(torch.equal(d_g.view(torch.int32), depth.view(torch.int32)) and torch.equal(b_g, bgr)))
eng.close()
