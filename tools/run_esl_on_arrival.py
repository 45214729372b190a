#!/usr/bin/env python3
"""The day the ESL data is there: BASELINE configs 0 / 2 and the reference's Table-1 row, in one command.

    python tools/run_esl_on_arrival.py --raw /ESL_data/static/seq1/data.raw --bias /ESL_data/static/seq1/data.bias \\
        --calib data/ESL_calib_hhi.yaml --scans /ESL_data/static/seq1/ --eval-calib data/calib.yaml --out report.json

Part A -- the replay (`.vscode/launch.json:24-49`: depth_reprojection.py --projector-width 1080 --projector-height 1920 --calib
data/ESL_calib_hhi.yaml --input data.raw --z-near 0.1 --z-far 1.2 --no-frame-dropping): the RAW file's words (EVT 3.0 or 2.0,
by its header) go through `with DepthReprojectionProcessor(params)` in chunks -- decoded on the device, polarity / activity
filter, trigger finder and the hot path there too (this build's default RuntimeParams) --; reported: frames shown, events, events/s,
ms per shown frame next to the reference's published 2.67 ms (Threadripper PRO 5955WX, frame stage only), and -- with
--compare-host-chain -- the same file through the opt-out host chain (NumPy decoder + filters + trigger finder), frame by frame.
The bias file only configures a live camera (bias_events_iterator.py:69-78): accepted, recorded, not used for a file.

Part B -- the accuracy row (eval/x-map-eval.sh:24-72 -> python/eval/compute_depth_x_maps.py:22-133 ->
python/eval/create_evaluation_table.py:84-180): every `<scans>/scans_np/*.npy` time surface -> X-maps depth in camera view on
the evaluation's tables (from_ESL_yaml, rect = 3 x projector, scan downwards) -> `<scans>/x_maps/depth_init/scansNNN.npy`
(+ point clouds as PLY with --point-clouds), then fill rate / RMSE against `<scans>/esl/depth_optim_filtered/*.npy` -- the
table's ground truth, which the reference's ESL baseline writes (out of scope here: without it the depth maps are written and
the row is reported as not computable) -- next to the published Book-Duck cell, FR 0.91 / RMSE 0.31 cm.

Tables come from the cv2-free builder (x_maps_amd/calibration.py: rectifying rotations pinned to OpenCV's on this calibration,
map rounding unpinned): a difference from the published numbers on the real data is therefore a finding about that builder or
the decoders, which is what this command exists to surface.  tests/test_gpu_on_arrival.py drives both parts on a recording this
build's own encoder writes; with the real files present (XM_ESL_DATA=/ESL_data/static/seq1) it runs them too."""
from __future__ import annotations

import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PUBLISHED = {"ms_per_frame": 2.67, "ms_per_frame_sd": 0.31, "hardware": "AMD Ryzen Threadripper PRO 5955WX",
             "seq1_book_duck": {"fill_rate": 0.91, "rmse_cm": 0.31}}


def raw_format(path: str) -> int:
    """3 or 2: the encoding a Prophesee RAW file's header declares"""
    from x_maps_amd import evt2, evt3
    with open(path, "rb") as f:
        head = f.read(1 << 16)
    fields, _ = evt3.split_raw_header(head)
    return 2 if evt2._is_evt2(fields) else 3


def replay_raw(raw, calib, proj_w, proj_h, fps, z_near, z_far, camera_perspective=False, device=0, chunk_words=1 << 20,
               device_ingest=True, keep_frames=0, tables=None, max_chunks=0):
    """Part A.  -> dict(report), list of (frame checksum, shape) per shown frame, the first `keep_frames` frames"""
    from x_maps_amd import evt2, evt3
    from x_maps_amd.depth_reprojection_processor import DepthReprojectionProcessor, RuntimeParams
    fmt = raw_format(raw)
    mod = evt2 if fmt == 2 else evt3
    shown, kept = [], []

    class Window:
        def should_close(self):
            return False

        def show_async(self, img):
            shown.append((img.shape, int(img[::7, ::5].astype(np.uint32).sum())))
            if len(kept) < keep_frames:
                kept.append(np.array(img))

    t_setup = time.perf_counter()
    params = RuntimeParams(camera_width=640, camera_height=480, projector_width=proj_w, projector_height=proj_h, projector_fps=fps,
                           z_near=z_near, z_far=z_far, calib=calib, projector_time_map=None, no_frame_dropping=True,
                           camera_perspective=camera_perspective, tables=tables, device=device, device_ingest=device_ingest)
    n_words = n_chunks = 0
    with DepthReprojectionProcessor(params, window=Window()) as proc:
        t_setup = time.perf_counter() - t_setup
        push = proc.process_evt2_words if fmt == 2 else proc.process_evt3_words
        c0 = time.perf_counter()
        for words in mod.read_raw_words(raw, chunk_words=chunk_words):
            push(words)
            n_words += len(words)
            n_chunks += 1
            if max_chunks and n_chunks >= max_chunks:
                break
        proc.flush()
        dt = time.perf_counter() - c0
        ds = proc._pipe.ingest.device_stats() if proc._pipe.ingest is not None else None
    rep = {"raw": raw, "format": f"EVT {fmt}.0", "words": n_words, "chunks": n_chunks, "chunk_words": chunk_words,
           "path": "device ingest (decode + filters + trigger finder + hot path on the GPU)" if device_ingest else
                   "host chain (NumPy decoder, filters, trigger finder; one fused GPU call per frame)",
           "setup_seconds": round(t_setup, 3), "replay_seconds": round(dt, 4), "frames_shown": len(shown),
           "ms_per_shown_frame": round(dt / max(len(shown), 1) * 1e3, 4), "reference_published_ms_per_frame": PUBLISHED["ms_per_frame"],
           "frame_shape": list(shown[0][0]) if shown else None}
    if ds is not None:
        rep["device"] = ds
        rep["events_behind_the_filters"] = ds["events_appended"]
        rep["Mevents_per_s_behind_the_filters"] = round(ds["events_appended"] / dt / 1e6, 2)
    return rep, shown, kept


def write_ply(path, pts):
    """binary little-endian PLY of an (k, 3) float32 point list (the reference writes its clouds through pyntcloud)"""
    pts = np.ascontiguousarray(pts, dtype="<f4")
    with open(path, "wb") as f:
        f.write((f"ply\nformat binary_little_endian 1.0\nelement vertex {len(pts)}\nproperty float x\nproperty float y\n"
                 f"property float z\nend_header\n").encode("ascii"))
        f.write(pts.tobytes())


def depth_from_scans(object_dir, eval_calib, proj_w, proj_h, num_scans=0, start_scan=0, point_clouds=False, device=0, tables=None):
    """Part B, first half: compute_depth_x_maps.py:22-133.  -> report"""
    from x_maps_amd import calibration as C
    from x_maps_amd.cam_proj_calibration import CamProjMaps
    from x_maps_amd.eval_depth import compute_depth_from_time_surface
    from x_maps_amd.x_maps_disparity import XMapsDisparity
    names = sorted(glob.glob(os.path.join(object_dir, "scans_np", "*.npy")))
    if not names:
        return {"error": f"no camera files found in {os.path.join(object_dir, 'scans_np')}"}
    depth_dir = os.path.join(object_dir, "x_maps", "depth_init")
    cloud_dir = os.path.join(object_dir, "x_maps", "pointcloud_init")
    os.makedirs(depth_dir, exist_ok=True)
    if point_clouds:
        os.makedirs(cloud_dir, exist_ok=True)
    t0 = time.perf_counter()
    if tables is None:
        cp = C.CamProjCalibrationParams.from_ESL_yaml(eval_calib, 640, 480, proj_w, proj_h)
        tables = C.build_eval_tables(cp, device=device)
    maps = CamProjMaps(tables, camera_perspective=True, device=device)
    xd = XMapsDisparity(maps)
    t_setup = time.perf_counter() - t0
    last = len(names) if not num_scans else min(len(names), start_scan + num_scans)
    per, skipped, filled = [], 0, []
    try:
        for i in range(start_scan, last):
            surf = np.load(names[i])
            c0 = time.perf_counter()
            depth, cloud = compute_depth_from_time_surface(maps, xd, surf, want_point_cloud=point_clouds)
            if depth is None:  # "Skip camera npy file ... since it is empty" (:132)
                skipped += 1
                continue
            per.append(time.perf_counter() - c0)
            np.save(os.path.join(depth_dir, "scans" + str(i).zfill(3) + ".npy"), depth)
            filled.append(float((depth > 0).mean()))
            if point_clouds:
                write_ply(os.path.join(cloud_dir, "scans" + str(i).zfill(3) + ".ply"), cloud)
    finally:
        maps.engine.close()
    return {"scans_found": len(names), "scans_processed": len(per), "scans_empty": skipped, "setup_seconds": round(t_setup, 3),
            "ms_per_scan_disparity_to_depth": round(float(np.mean(per)) * 1e3, 4) if per else None,
            "mean_fraction_of_pixels_with_depth": round(float(np.mean(filled)), 4) if filled else None,
            "reference_published_ms_per_frame": PUBLISHED["ms_per_frame"], "depth_dir": depth_dir}


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--raw", help="Prophesee RAW recording (EVT 3.0 / 2.0), e.g. /ESL_data/static/seq1/data.raw")
    ap.add_argument("--bias", help="camera bias file of the recording (recorded in the report; a file replay does not use it)")
    ap.add_argument("--calib", default=os.path.join(ROOT, "..", "reference", "data", "ESL_calib_hhi.yaml"),
                    help="calibration YAML of the live pipe (the reference's data/ESL_calib_hhi.yaml)")
    ap.add_argument("--scans", help="sequence directory holding scans_np/*.npy (and, for the table row, esl/depth_optim_filtered/*.npy)")
    ap.add_argument("--eval-calib", help="the ESL dataset's calib.yaml (cam_K, cam_kc, proj_K, proj_kc, R, T) for part B")
    ap.add_argument("--projector-width", type=int, default=1080)
    ap.add_argument("--projector-height", type=int, default=1920)
    ap.add_argument("--projector-fps", type=int, default=60)
    ap.add_argument("--z-near", type=float, default=0.1)
    ap.add_argument("--z-far", type=float, default=1.2)
    ap.add_argument("--camera-perspective", action="store_true")
    ap.add_argument("--compare-host-chain", action="store_true", help="part A a second time through the opt-out host chain, frames compared")
    ap.add_argument("--chunk-words", type=int, default=1 << 20)
    ap.add_argument("--num-scans", type=int, default=0, help="part B: scans to process (0 = all)")
    ap.add_argument("--start-scan", type=int, default=0)
    ap.add_argument("--point-clouds", action="store_true")
    ap.add_argument("--min-depth", type=float, default=20)
    ap.add_argument("--max-depth", type=float, default=500, help="eval/x-map-eval.sh:72 passes 500 (the table script's own default is 120)")
    ap.add_argument("--save-frames", type=int, default=0, help="part A: keep the first N BGR frames as frame_NNN.npy beside --out")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--out", help="write the JSON report here as well")
    a = ap.parse_args(argv)
    if not a.raw and not a.scans:
        ap.error("nothing to do: give --raw and / or --scans")
    report = {"published": PUBLISHED}
    if a.raw:
        if a.bias:
            report["bias_file"] = {"path": a.bias, "present": os.path.exists(a.bias), "note": "configures a live camera only"}
        rep, shown, kept = replay_raw(a.raw, a.calib, a.projector_width, a.projector_height, a.projector_fps, a.z_near, a.z_far,
                                      a.camera_perspective, a.device, a.chunk_words, True, a.save_frames)
        report["replay"] = rep
        if a.save_frames and a.out:
            for i, f in enumerate(kept):
                np.save(os.path.join(os.path.dirname(os.path.abspath(a.out)), f"frame_{i:03d}.npy"), f)
        if a.compare_host_chain:
            rep_h, shown_h, _ = replay_raw(a.raw, a.calib, a.projector_width, a.projector_height, a.projector_fps, a.z_near, a.z_far,
                                           a.camera_perspective, a.device, a.chunk_words, False, 0)
            report["replay_host_chain"] = rep_h
            report["replay"]["same_frames_as_host_chain"] = bool(shown == shown_h)
    if a.scans:
        if not a.eval_calib:
            ap.error("--scans needs --eval-calib (the ESL dataset's calib.yaml)")
        report["depth_from_scans"] = depth_from_scans(a.scans, a.eval_calib, a.projector_width, a.projector_height, a.num_scans,
                                                      a.start_scan, a.point_clouds, a.device)
        if "error" not in report["depth_from_scans"]:
            from x_maps_amd.eval_table import x_maps_table_row
            row = x_maps_table_row(a.scans, a.min_depth, a.max_depth, device=a.device)
            row["published_cell_seq1"] = "{fill_rate} & {rmse_cm}".format(**PUBLISHED["seq1_book_duck"])
            report["table_1_row_x_maps"] = row
    txt = json.dumps(report, indent=1)
    print(txt)
    if a.out:
        with open(a.out, "w") as f:
            f.write(txt + "\n")
    return report


if __name__ == "__main__":
    main()
