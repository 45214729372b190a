"""-m gpu: A4 (cv2.dilate 7x7 o cv2.remap nearest, python/disp_to_depth.py:76-97) on the GPU against a brute-force
49-tap DEFINITION written out here (not the oracle's restatement), on border-heavy maps: targets within 3 px of every
frame edge, map entries outside the frame on all four sides, odd and even frame heights, tiles that straddle the frame.

Both implementations are checked: the stage kernel (xm_stage_remap_rectified_disp_map_to_proj, f32 frame in) and the
fused tiled frame kernel K2 (through xm_shard_finish on a hand-built packed-key frame, so the frame content is arbitrary
and reaches every border cell -- the last rectified row can never be written by K1 itself, xmd:23).

Semantics restated from the OpenCV 4.x documentation (the reference's environment pins no version: conda-forge `opencv`,
py3.8 era => 4.5-4.8; .devcontainer/environment.yaml:18): cv2.dilate with a 7x7 all-ones kernel, default anchor = centre,
default borderType BORDER_CONSTANT with borderValue = morphologyDefaultBorderValue() = "-inf for dilation", i.e. cells
outside the image never win the max; cv2.remap(INTER_NEAREST, map1 = CV_16SC2 (x, y), borderMode = BORDER_CONSTANT,
borderValue = 0): dst(v, u) = src(my, mx) when inside, else 0.
"""
import numpy as np
import pytest

import xmaps_oracle as O
from x_maps_amd import XMapsEngine
from x_maps_amd import synthetic as S

pytestmark = pytest.mark.gpu

KEY_IDX_SHIFT, KEY_TAG_SHIFT = 16, 44


def brute_dilate_remap(rect: np.ndarray, pmap: np.ndarray) -> np.ndarray:
    """49 taps per output pixel, straight from the definition."""
    H, W = rect.shape
    ph, pw = pmap.shape[:2]
    out = np.zeros((ph, pw), np.float32)
    for v in range(ph):
        for u in range(pw):
            mx, my = int(pmap[v, u, 0]), int(pmap[v, u, 1])
            if not (0 <= mx < W and 0 <= my < H):
                continue  # BORDER_CONSTANT 0
            best = -np.inf
            for dy in range(-3, 4):
                for dx in range(-3, 4):
                    yy, xx = my + dy, mx + dx
                    if 0 <= yy < H and 0 <= xx < W:
                        best = max(best, float(rect[yy, xx]))
            out[v, u] = best
    return out


def border_tables(rect_w, rect_h, proj_w, proj_h, seed):
    """Tables whose projector map sweeps from 6 px outside the rectified frame on one side to 6 px outside on the other,
    with every pixel's target jittered -- so targets sit 0,1,2,3 px from each edge and beyond it."""
    rng = np.random.default_rng(seed)
    tb = S.make_tables(S.C_TINY)
    vs, us = np.mgrid[0:proj_h, 0:proj_w].astype(np.float64)
    mx = np.rint(-6 + us * (rect_w + 12) / max(proj_w - 1, 1) + rng.integers(-2, 3, us.shape))
    my = np.rint(-6 + vs * (rect_h + 12) / max(proj_h - 1, 1) + rng.integers(-2, 3, vs.shape))
    # pin a few targets exactly onto the corners / edges
    mx[0, :4] = [0, 1, 2, 3]
    my[0, :4] = [0, 0, 0, 0]
    mx[-1, -4:] = [rect_w - 4, rect_w - 3, rect_w - 2, rect_w - 1]
    my[-1, -4:] = rect_h - 1
    mx[1, :3] = [-1, rect_w, 5]
    my[1, :3] = [5, 5, rect_h]
    tb.update({"rect_w": rect_w, "rect_h": rect_h, "proj_w": proj_w, "proj_h": proj_h,
               "disp_proj_mapxy_i16": np.ascontiguousarray(np.stack((mx, my), -1).astype(np.int16)),
               "proj_x_map": np.zeros((rect_h, tb["proj_x_map"].shape[1]), np.int16)})
    return tb, rng


def sparse_frame(rng, rect_w, rect_h, fill):
    rect = rng.integers(1, 900, (rect_h, rect_w)).astype(np.float32)
    rect[rng.random(rect.shape) >= fill] = 0
    # make sure the border cells themselves carry values (they decide the edge cases)
    rect[0, :] = rng.integers(1, 900, rect_w)
    rect[-1, :] = rng.integers(1, 900, rect_w)
    rect[:, 0] = rng.integers(1, 900, rect_h)
    rect[:, -1] = rng.integers(1, 900, rect_h)
    return rect


CASES = [(176, 132, 64, 48, 0.05), (151, 101, 50, 37, 0.3), (97, 64, 33, 70, 0.02), (40, 23, 19, 17, 0.5),
         # projector widths around K2's 32-pixel tile (two pixels per thread: columns tx and tx + 16), heights around its 16 rows
         (60, 44, 16, 9, 0.3), (60, 44, 17, 16, 0.3), (66, 40, 31, 20, 0.2), (90, 48, 32, 16, 0.2), (90, 48, 96, 5, 0.1),
         (30, 20, 4, 5, 0.5), (352, 264, 128, 96, 0.4)]


@pytest.mark.parametrize("rect_w,rect_h,proj_w,proj_h,fill", CASES)
def test_stage_a4_equals_the_49_tap_definition(rect_w, rect_h, proj_w, proj_h, fill):
    tb, rng = border_tables(rect_w, rect_h, proj_w, proj_h, seed=rect_w)
    rect = sparse_frame(rng, rect_w, rect_h, fill)
    want = brute_dilate_remap(rect, tb["disp_proj_mapxy_i16"])
    with XMapsEngine(tb) as eng:
        got = eng.remap_rectified_disp_map_to_proj(rect)
    assert np.array_equal(got, want)
    assert np.array_equal(O.remap_rectified_disp_map_to_proj(rect, tb["disp_proj_mapxy_i16"]), want)  # the oracle too


@pytest.mark.parametrize("rect_w,rect_h,proj_w,proj_h,fill", CASES)
def test_fused_k2_equals_the_49_tap_definition(rect_w, rect_h, proj_w, proj_h, fill):
    """The tiled frame kernel of the fused path (LDS patches, separable max, per-tile tables) on an arbitrary key frame."""
    torch = pytest.importorskip("torch")
    tb, rng = border_tables(rect_w, rect_h, proj_w, proj_h, seed=1000 + rect_w)
    rect = sparse_frame(rng, rect_w, rect_h, fill)
    tag = 7
    idx = rng.integers(0, 1 << 20, rect.shape).astype(np.uint64)
    keys = (np.uint64(tag) << np.uint64(KEY_TAG_SHIFT)) | (idx << np.uint64(KEY_IDX_SHIFT)) | rect.astype(np.uint64)
    keys[rect == 0] = 0
    # stale cells of an older frame (smaller tag) must read as empty
    stale = (rect == 0) & (rng.random(rect.shape) < 0.3)
    keys[stale] = (np.uint64(tag - 1) << np.uint64(KEY_TAG_SHIFT)) | np.uint64(777)
    want_disp = brute_dilate_remap(rect, tb["disp_proj_mapxy_i16"])
    want_depth = O.disparity_to_depth_rectified(want_disp, tb["p03"])
    dev = torch.device("cuda", 0)
    with XMapsEngine(tb) as eng:
        # projector-view key frame is column-major [col][row]
        kf = torch.from_numpy(np.ascontiguousarray(keys.T).view(np.int64)).to(dev)
        depth = torch.zeros((proj_h, proj_w), dtype=torch.float32, device=dev)
        bgr = torch.zeros((proj_h, proj_w, 3), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        eng.shard_finish(kf.data_ptr(), tag, depth.data_ptr(), bgr.data_ptr())
        eng.sync()
        got = depth.cpu().numpy()
        got_bgr = bgr.cpu().numpy()
    assert np.array_equal(got, want_depth)
    u8 = O.clip_normalize_uint8_depth_frame(want_depth, tb["z_near"], tb["z_far"])
    assert np.array_equal(got_bgr, O.generate_color_map(u8))


@pytest.mark.parametrize("rect_w,rect_h,proj_w,proj_h,fill", CASES)
def test_fused_k2_on_a_plain_u16_frame_equals_the_49_tap_definition(rect_w, rect_h, proj_w, proj_h, fill):
    """The same kernel on the untagged 2-byte disparity frame (what the column tiles and the reduce-scatter merge hand it):
    16-byte loads of 8 rows when rect_h % 8 == 0, 8-byte loads of 4 rows when % 4, cell by cell otherwise and along the border."""
    torch = pytest.importorskip("torch")
    tb, rng = border_tables(rect_w, rect_h, proj_w, proj_h, seed=2000 + rect_w)
    rect = sparse_frame(rng, rect_w, rect_h, fill)
    want_disp = brute_dilate_remap(rect, tb["disp_proj_mapxy_i16"])
    want_depth = O.disparity_to_depth_rectified(want_disp, tb["p03"])
    dev = torch.device("cuda", 0)
    with XMapsEngine(tb) as eng:
        d16 = torch.from_numpy(np.ascontiguousarray(rect.T).astype(np.uint16).view(np.int16)).to(dev)  # column-major [col][row]
        depth = torch.zeros((proj_h, proj_w), dtype=torch.float32, device=dev)
        bgr = torch.zeros((proj_h, proj_w, 3), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        eng.shard_finish_u16(d16.data_ptr(), depth.data_ptr(), bgr.data_ptr())
        eng.sync()
        got = depth.cpu().numpy()
        got_bgr = bgr.cpu().numpy()
    assert np.array_equal(got, want_depth)
    u8 = O.clip_normalize_uint8_depth_frame(want_depth, tb["z_near"], tb["z_far"])
    assert np.array_equal(got_bgr, O.generate_color_map(u8))
