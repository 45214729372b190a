import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import xmaps_oracle as O
from x_maps_amd import XMapsEngine, synthetic as S
from x_maps_amd.ingest import DeviceIngest
camera = os.environ.get("CAM", "1") == "1"
cfg = S.C_1M
tb = S.make_tables(cfg)
n_ev = cfg.n_events
host_frames = [S.to_soa(S.make_events(cfg, frame=f))[:3] for f in range(4)]
n_frames = 6
period = 16_600
with XMapsEngine(tb, camera_perspective=camera, n_slots=4) as eng:
    stream = eng.host_empty((n_frames * n_ev,), S.EVENT_CD_DTYPE)
    for f in range(n_frames):
        x, y, t = host_frames[f % 4]
        sl = stream[f * n_ev:(f + 1) * n_ev]
        sl["x"], sl["y"], sl["p"] = x, y, 1
        sl["t"] = t - t[0] + 2_000_000 + f * period
    packet = int(1e6 / 60 / 4)
    edges = np.arange(stream["t"][0], stream["t"][-1] + packet, packet)
    cuts = np.searchsorted(stream["t"], edges)
    with DeviceIngest(eng, 60, capacity_events=1 << 22, max_packet_events=1 << 20, expected_events_per_frame=n_ev, result_ring=8) as ing:
        for a, b in zip(cuts[:-1], cuts[1:]):
            ing.push_pinned(stream[a:b])
        ing.flush()
        got = ing.poll()
    print("frames", len(got))
    for f0 in got:
        i0 = int(np.searchsorted(stream["t"], f0.t_first))
        ev0 = stream[i0:i0 + f0.n_events]
        ref = O.process_ev_frame(tb, ev0["x"].astype(np.int64), ev0["y"].astype(np.int64), np.ascontiguousarray(ev0["t"]), camera_perspective=camera, want_bgr=True)
        d = f0.depth
        print(f0.t_first, f0.n_events, "shape", d.shape, ref["depth"].shape, "equal", np.array_equal(d, ref["depth"]), "ndiff", int((d != ref["depth"]).sum()),
              "bgr equal", np.array_equal(f0.bgr, ref["bgr"]), "t sorted", bool((np.diff(ev0["t"]) >= 0).all()), "t_last", int(ev0["t"][-1]), getattr(f0, "t_last", None))
        if not np.array_equal(d, ref["depth"]):
            ys, xs = np.nonzero(d != ref["depth"])
            print("  first diffs", list(zip(ys[:5], xs[:5])), d[ys[:5], xs[:5]], ref["depth"][ys[:5], xs[:5]])
            # same frame through the engine's single-frame call
            d1, b1, st = eng.process_events(ev0)
            print("  single-frame call equal oracle:", np.array_equal(d1, ref["depth"]))
    # ---- which configuration reproduces the ingest's result?
    f0 = got[0]
    i0 = int(np.searchsorted(stream["t"], f0.t_first))
    ev0 = np.array(stream[i0:i0 + f0.n_events])
    ref = O.process_ev_frame(tb, ev0["x"].astype(np.int64), ev0["y"].astype(np.int64), np.ascontiguousarray(ev0["t"]), camera_perspective=camera)
    sel = (ev0["y"] == 401) & (ev0["x"] == 639)
    print("events on (401, 639):", np.nonzero(sel)[0], ev0["t"][sel])
    x_, y_, t_, _ = S.to_soa(ev0)
    dbg = eng.debug_event_outputs(x_, y_, t_)
    idx = np.nonzero(sel)[0]
    print("disp", dbg["disp"][idx], "mask", dbg["mask"][idx], "depth each", [float(np.float32(tb["p03"] / d)) if d else 0 for d in dbg["disp"][idx]])
    print("ingest stats", f0.n_events, getattr(f0, "n_inliers", None), "oracle inliers", int(ref["mask"].sum()))
    # the four events the trigger finder trims (2 in front, 2 behind)
    for k in (i0 - 2, i0 - 1, i0 + f0.n_events, i0 + f0.n_events + 1):
        if 0 <= k < len(stream): print("  trimmed", k - i0, stream[k])
