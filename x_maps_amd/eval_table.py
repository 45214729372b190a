"""The X-maps row of the reference's Table 1, from files on disk (python/eval/create_evaluation_table.py:57-62,84-180 and the
ground-truth combination it borrows from python/eval/esl_utilities.py:153-175), for one sequence directory laid out as the
reference's evaluation leaves it:

    <seq>/scans_np/*.npy                   camera time surfaces (input of compute_depth_x_maps.py)
    <seq>/x_maps/depth_init/scansNNN.npy   X-maps depth per scan        (written by x_maps_amd / tools/run_esl_on_arrival.py)
    <seq>/esl/depth_optim_filtered/*.npy   ESL's optimised depth = the table's ground truth (written by the reference's
                                           compute_depth_esl.py -- a competitor algorithm, out of scope here: when the directory
                                           is absent the row cannot be computed and the caller says so)

The metrics themselves run on the GPU (x_maps_amd.eval_metrics, pinned by golden G8); what is restated here is the file-level
recipe: the per-pixel mean of the filtered ground-truth scans, median-filtered 3 x 3 (cv2.medianBlur: unpinned against a cv2
run -- OpenCV is not in this image -- replicate border, exact median of nine), as the mask every map is filtered with."""
from __future__ import annotations

import glob
import os

import numpy as np

from .eval_metrics import evaluation_stats


def median_blur3(img: np.ndarray) -> np.ndarray:
    """cv2.medianBlur(img, 3) for a float32 image: median of the 3 x 3 neighbourhood, borders replicated"""
    p = np.pad(np.asarray(img, np.float32), 1, mode="edge")
    h, w = img.shape
    stack = np.stack([p[dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3)], 0)
    return np.partition(stack, 4, axis=0)[4]


def combine_depth_maps(depth_files, min_d, max_d):
    """esl_utilities.py:153-175 (combine_mc3d): mean over the scans of the depths inside (min_d, max_d), median-filtered;
    -> (combined, 1 % of the mean depth, mean depth)"""
    acc = count = None
    for f in depth_files:
        try:
            d = np.load(f).astype(np.float32)
        except Exception:  # (the reference skips unreadable files too)
            continue
        if acc is None:
            acc, count = np.zeros(d.shape, np.float32), np.zeros(d.shape, np.float32)
        d[d >= max_d] = 0
        d[d <= min_d] = 0
        acc += d
        count += d > 0
    if acc is None:
        raise ValueError("no readable depth map")
    with np.errstate(invalid="ignore", divide="ignore"):
        comb = acc / count
    comb[count == 0] = 0
    comb = median_blur3(comb)
    avg = float(comb[comb > 0].sum() / max(int((comb > 0).sum()), 1))
    return comb, 0.01 * avg, avg


def load_and_filter(filename, gt, min_depth, max_depth):
    """create_evaluation_table.py:57-62"""
    r = np.load(filename).astype(np.float32)
    r[r >= max_depth] = 0
    r[r <= min_depth] = 0
    r[gt == 0] = 0
    return r


def x_maps_table_row(seq_dir: str, min_depth: float = 20, max_depth: float = 120, device: int = 0) -> dict:
    """Mean fill rate / RMSE of <seq>/x_maps/depth_init against <seq>/esl/depth_optim_filtered, scan by scan
    (create_evaluation_table.py:121-160; its defaults are min 20 / max 120, eval/x-map-eval.sh:72 passes -max_depth 500)."""
    gt_files = sorted(glob.glob(os.path.join(seq_dir, "esl", "depth_optim_filtered", "*.npy")))
    xm_files = sorted(glob.glob(os.path.join(seq_dir, "x_maps", "depth_init", "*.npy")))
    if not gt_files:
        return {"error": f"no ground truth under {os.path.join(seq_dir, 'esl', 'depth_optim_filtered')} (the reference's "
                         f"compute_depth_esl.py writes it; that baseline is out of scope here)", "x_maps_files": len(xm_files)}
    if len(gt_files) != len(xm_files):
        return {"error": "frames are missing", "gt_files": len(gt_files), "x_maps_files": len(xm_files)}
    gt_combined, _, avg_depth = combine_depth_maps(gt_files, min_depth, max_depth)
    rows = []
    for g, x in zip(gt_files, xm_files):
        gt = load_and_filter(g, gt_combined, min_depth, max_depth)
        est = load_and_filter(x, gt_combined, min_depth, max_depth)
        s = evaluation_stats(est, gt, device=device)
        rows.append((s.fillrate, s.rmse))
    fr, rmse = np.mean(np.array(rows, np.float64), axis=0)
    return {"scans": len(rows), "mean_depth": round(avg_depth, 3), "fill_rate": float(fr), "rmse": float(rmse),
            "table_cell": f"{round(float(fr), 2)} & {round(float(rmse), 2)}"}
