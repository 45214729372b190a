#!/usr/bin/env python3
"""Experiment: ESL-like rig end to end on the GPU: parity vs oracle, accuracy vs ground truth, kernel times."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import xmaps_oracle as O
from x_maps_amd import XMapsEngine, rig, synthetic as S
t0 = time.time()
cp, tb, evs, gt = rig.make_esl_like(row_stride=int(os.environ.get("STRIDE", "13")))
print(f"tables+events in {time.time()-t0:.2f}s: n={len(evs)}  rect {tb['rect_w']}x{tb['rect_h']} xmap {tb['proj_x_map'].shape} p03={tb['p03']:.3f} f={tb['P1'][0,0]:.1f}")
print("xmap defined frac", (tb['proj_x_map']>0).mean(), " time_map_rect nonzero", (tb['time_map_rect']>0).mean())
x, y, t, _ = S.to_soa(evs)
ref = O.process_ev_frame(tb, x.astype(np.int64), y.astype(np.int64), t)
print("oracle inliers", ref['mask'].sum(), "disp range", ref['disp'].min(), ref['disp'].max(), "proj filled", (ref['depth']>0).mean())
with XMapsEngine(tb, n_slots=1) as eng:
    depth, bgr, st = eng.process_events(evs)
    print("GPU == oracle depth:", np.array_equal(depth, ref['depth']), " bgr:", np.array_equal(bgr, ref['bgr']), st.n_inliers)
    est = depth[gt['proj_v'], gt['proj_u']]
    ok = est > 0
    rel = np.abs(est[ok] - gt['z_rect'][ok]) / gt['z_rect'][ok]
    print(f"accuracy: fill {ok.mean():.3f}  median rel err {np.median(rel)*100:.3f}%  p95 {np.percentile(rel,95)*100:.2f}%  RMSE {np.sqrt(np.mean((est[ok]-gt['z_rect'][ok])**2))*100:.3f} cm")
    dev = torch.device("cuda", 0)
    X, Y, T = (torch.from_numpy(a).to(dev) for a in (x.view(np.int16), y.view(np.int16), t))
    d = torch.empty((eng.out_h, eng.out_w), dtype=torch.float32, device=dev); b = torch.empty((eng.out_h, eng.out_w, 3), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    ms = np.array([eng.profile_frame_device(X.data_ptr(), Y.data_ptr(), T.data_ptr(), None, len(t), d.data_ptr(), b.data_ptr()).gpu_ms for _ in range(40)])[10:].mean(0) * 1e3
    print(f"kernel us: minmax {ms[0]:.2f} scatter {ms[1]:.2f} frame {ms[2]:.2f} total {ms[3]:.2f}")
