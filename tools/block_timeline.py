#!/usr/bin/env python3
"""Experiment (needs the -DXM_BLOG build, see tools/block_timeline.sh): block-level GPU timeline of the pipelined frame
loop.  Every block of K0/K1/K2 logs its start/end in the 100 MHz real-time counter; from that: the window of every launch,
how many launches / blocks run at once, and what a frame's critical path looks like."""
import collections, ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x_maps_amd import XMapsEngine, synthetic as S, _native as N
dev = torch.device("cuda", 0)
cfg = S.C_1M
tb = S.make_tables(cfg)
slots = int(os.environ.get("SLOTS", "8"))
sorted_mode = bool(int(os.environ.get("SORTED", "0")))
frames = []
for f in range(8):
    x, y, t, _ = S.to_soa(S.make_events(cfg, frame=f))
    frames.append(tuple(torch.from_numpy(a).to(dev) for a in (x.view(np.int16), y.view(np.int16), t)))
eng = XMapsEngine(tb, n_slots=slots, assume_time_sorted=sorted_mode)
out = torch.empty((slots, cfg.proj_h, cfg.proj_w), dtype=torch.float32, device=dev)
bgr = torch.empty((slots, cfg.proj_h, cfg.proj_w, 3), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
lib = N.load_library()
lib.xm_debug_blog.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.POINTER(ctypes.c_uint)]
def run(n):
    for i in range(n):
        fx, fy, ft = frames[i % 8]
        eng.process_frame_device(fx.data_ptr(), fy.data_ptr(), ft.data_ptr(), None, cfg.n_events,
                                 out[i % slots].data_ptr(), bgr[i % slots].data_ptr())
    eng.sync()
run(40)
cnt = ctypes.c_uint(0)
lib.xm_debug_blog(None, 0, ctypes.byref(cnt))  # clear
K = 8 * slots if slots > 1 else 8   # the log keeps the last 8 frames of every slot
t0 = time.perf_counter(); run(K); dt = time.perf_counter() - t0
cap = 8 * 16 * 4096
buf = np.zeros((cap, 4), np.uint64)
lib.xm_debug_blog(buf.ctypes.data, cap, ctypes.byref(cnt))
sel = np.nonzero((buf[:, 0] >> np.uint64(63)) == 1)[0]
b = buf[sel]
blk = (sel % 4096) % 1024 if False else (sel % 4096)
n = len(b)
kind = (b[:, 0] & np.uint64(0xff)).astype(np.int64); tag = ((b[:, 0] >> np.uint64(8)) & np.uint64(0xffffffff)).astype(np.int64)
st = ((b[:, 0] >> np.uint64(40)) & np.uint64(0xff)).astype(np.int64)
ts = b[:, 1].astype(np.int64); te = b[:, 2].astype(np.int64)
T0 = ts.min(); ts -= T0; te -= T0
tick_us = 0.01  # 100 MHz
print(f"slots {slots} sorted {sorted_mode}: host loop {dt/K*1e6:.2f} us/frame; {n} block records over {(te.max())*tick_us:.0f} us -> {(te.max())*tick_us/K:.2f} us/frame on the GPU clock")
launch = collections.defaultdict(lambda: [1 << 62, 0, 0, 0])
for k, g, s_, a, e in zip(kind, tag, st, ts, te):
    L = launch[(int(s_), int(g), int(k))]
    L[0] = min(L[0], a); L[1] = max(L[1], e); L[2] += 1; L[3] += e - a
names = {0: "K0", 1: "K1", 2: "K2"}
for k in (0, 1, 2):
    Ls = [v for (s_, g, kk), v in launch.items() if kk == k]
    if not Ls: continue
    w = np.array([(v[1] - v[0]) * tick_us for v in Ls]); nb = np.array([v[2] for v in Ls]); bt = np.array([v[3] / v[2] * tick_us for v in Ls])
    print(f"{names[k]}: {len(Ls)} launches, window mean {w.mean():6.2f} us (p10 {np.percentile(w,10):.2f}, p90 {np.percentile(w,90):.2f}), {nb.mean():.0f} blocks, mean block life {bt.mean():.2f} us")
# inside a launch: when do its blocks start (dispatch skew) and how long do they live?
for k in (0, 1, 2):
    offs, lifes = [], []
    for (s_, g, kk), v in launch.items():
        if kk != k: continue
        m = (kind == k) & (st == s_) & (tag == g)
        offs.append((ts[m] - v[0]) * tick_us); lifes.append((te[m] - ts[m]) * tick_us)
    if offs:
        o = np.concatenate(offs); l = np.concatenate(lifes)
        print(f"   {names[k]} block start offset in its launch: p50 {np.percentile(o,50):5.2f} p90 {np.percentile(o,90):5.2f} max {o.max():5.2f} us | block life: p10 {np.percentile(l,10):5.2f} p50 {np.percentile(l,50):5.2f} p90 {np.percentile(l,90):5.2f} max {l.max():5.2f} us")
# concurrency of launches over time (steady part)
lo, hi = np.percentile(ts, 15), np.percentile(te, 85)
ev = []
for (s_, g, k), v in launch.items():
    ev.append((v[0], 1, k)); ev.append((v[1], -1, k))
ev.sort()
cur = collections.Counter(); hist = collections.Counter(); last = None
for t, d, k in ev:
    if last is not None and t > lo and last < hi:
        a, b_ = max(last, lo), min(t, hi)
        if b_ > a: hist[(cur[0], cur[1], cur[2])] += b_ - a
    cur[k] += d; last = t
tot = sum(hist.values())
print("launches in flight (K0,K1,K2) -> share of time:")
for key, v in sorted(hist.items(), key=lambda kv: -kv[1])[:12]:
    print(f"   {key}: {100*v/tot:5.1f} %")
avg = [sum(key[i] * v for key, v in hist.items()) / tot for i in range(3)]
print(f"   mean in flight: K0 {avg[0]:.2f}  K1 {avg[1]:.2f}  K2 {avg[2]:.2f}   idle (nothing running) {100*hist.get((0,0,0),0)/tot:.1f} %")
# per-slot chain: K0 start -> K1 start -> K2 start -> K2 end -> next K0 start
chains = collections.defaultdict(list)
for (s_, g, k), v in launch.items(): chains[s_].append((v[0], v[1], k, g))
gaps = collections.defaultdict(list)
for s_, lst in chains.items():
    lst.sort()
    for (a0, e0, k0, g0), (a1, e1, k1, g1) in zip(lst, lst[1:]):
        if lo < a0 < hi: gaps[f"{names[k0]} end -> {names[k1]} start"].append((a1 - e0) * tick_us)
for k, v in sorted(gaps.items()):
    v = np.array(v); print(f"   gap {k}: mean {v.mean():6.2f} us  p10 {np.percentile(v,10):6.2f}  p90 {np.percentile(v,90):6.2f}  (n={len(v)})")
# blocks resident over time
bev = np.concatenate([np.stack([ts, np.ones_like(ts), kind], 1), np.stack([te, -np.ones_like(te), kind], 1)])
bev = bev[np.argsort(bev[:, 0], kind="stable")]
res = np.zeros(3); acc = np.zeros(3); last = bev[0, 0]
for t, d, k in bev:
    if t > lo and last < hi: acc += res * (min(t, hi) - max(last, lo))
    res[int(k)] += d; last = t
print("   mean resident blocks: " + "  ".join(f"{names[i]} {acc[i]/(hi-lo):7.1f}" for i in range(3)))

if slots == 1:
    m = kind == 1
    lb = (blk[m] - 1024); life = (te[m] - ts[m]) * tick_us
    agg = collections.defaultdict(list)
    for i, l in zip(lb, life): agg[int(i)].append(l)
    mean = {i: np.mean(v) for i, v in agg.items()}
    order = sorted(mean, key=lambda i: -mean[i])
    print("   K1 slowest blocks (blockIdx: mean life us): " + "  ".join(f"{i}:{mean[i]:.1f}" for i in order[:24]))
    print("   K1 fastest blocks: " + "  ".join(f"{i}:{mean[i]:.1f}" for i in order[-8:]))
    byx = collections.defaultdict(list)
    for i, v in mean.items(): byx[i % 8].append(v)
    print("   K1 mean life by blockIdx % 8 (XCD): " + "  ".join(f"{x}:{np.mean(v):.2f}" for x, v in sorted(byx.items())))
