"""CPU: the cv2-free table builder (x_maps_amd/calibration.py, row N4).  stereo_rectify is pinned against everything OpenCV's
stereoRectify wrote into the reference's calibration file (R1, R2, P1, P2, Q, both ROIs: bit for bit); the maps are checked for
geometric consistency."""
import os

import numpy as np

from x_maps_amd import calibration as C


def _g(golden_dir):
    return np.load(os.path.join(golden_dir, "g6_esl_calib.npz"))


def test_stereo_rectify_reproduces_the_opencv_outputs_stored_in_the_reference_yaml(golden_dir):
    """data/ESL_calib_hhi.yaml:62-134 holds what a real cv::stereoRectify returned for this rig: R1, R2, P1, P2, Q and both
    validPixROIs, computed with alpha = 0.5 (:138).  The call that produced them -- camera first, imageSize = (480, 640) (the
    file's img_shape, rows first), newImageSize = (1920, 1080) (proj_shape), T in centimetres, CALIB_ZERO_DISPARITY -- is
    reproduced here to the last digit: the rotations, the common focal length (mean of the two, x newImageSize / imageSize),
    the principal points (float32 corner images), the inner / outer rectangles behind the alpha scaling and the ROIs.  The
    alpha = -1 path the reference takes (python/cam_proj_calibration.py:203-217) shares all of it but the scaling."""
    g = _g(golden_dir)
    R1, R2, P1, P2, Q, roi1, roi2 = C.stereo_rectify(g["camera_K"], g["camera_D"], g["projector_K"], g["projector_D"], (480, 640),
                                                     g["R"], g["T"].reshape(3) * 100.0, alpha=0.5, new_image_size=(1920, 1080),
                                                     return_rois=True)
    assert np.abs(R1 - g["R1"]).max() < 1e-12 and np.abs(R2 - g["R2"]).max() < 1e-12
    for got, want in ((P1, g["P1"]), (P2, g["P2"]), (Q, g["Q"])):
        assert np.abs(got - want).max() <= 1e-9 * max(1.0, np.abs(want).max()), (got, want)
    assert roi1 == (0, 0, 1920, 1080) and roi2 == (0, 0, 0, 0)  # validPixROI1 / validPixROI2 of the file (:118-133)
    assert np.allclose(R1 @ R1.T, np.eye(3), atol=1e-12) and abs(np.linalg.det(R1) - 1) < 1e-12
    # the alpha = -1 call of the reference: same rotations, no scaling, CALIB_ZERO_DISPARITY
    r1, r2, p1, p2, q = C.stereo_rectify(g["camera_K"], g["camera_D"], g["projector_K"], g["projector_D"], (1920, 1080), g["R"], g["T"])
    assert np.abs(r1 - g["R1"]).max() < 1e-12 and np.abs(r2 - g["R2"]).max() < 1e-12
    assert p1[0, 0] == p1[1, 1] == p2[0, 0] == (g["camera_K"][1, 1] + g["projector_K"][1, 1]) / 2
    assert p1[0, 3] == 0 and p2[0, 3] != 0 and p2[1, 3] == 0 and np.array_equal(p1[:, :3], p2[:, :3])


def test_rodrigues_round_trip():
    rng = np.random.default_rng(0)
    for _ in range(20):
        v = rng.normal(0, 1, 3)
        v *= rng.uniform(0.01, 3.0) / np.linalg.norm(v)
        assert np.allclose(C.rodrigues(C.rodrigues(v)), v, atol=1e-9)
    assert np.allclose(C.rodrigues(np.zeros(3)), np.eye(3))


def test_rectification_aligns_epipolar_lines_and_maps_invert(golden_dir):
    """Random 3-D points seen by both devices land on the same rectified row (within the rounding of the maps) and the
    inverse map undoes the forward map."""
    g = _g(golden_dir)
    D_cam = np.array([-6.4e-4, 5.9e-3, -1e-4, 4.3e-4, 0.12])
    size = (1760, 1320)
    R1, R2, P1, P2, Q = C.stereo_rectify(g["projector_K"], np.zeros(5), g["camera_K"], D_cam, size, g["R"], g["T"])
    rng = np.random.default_rng(1)
    xyz_c = np.stack((rng.uniform(-0.15, 0.15, 500), rng.uniform(-0.1, 0.1, 500), rng.uniform(0.4, 0.9, 500)), -1)
    xyz_p = xyz_c @ g["R"].T + g["T"].reshape(1, 3)
    pc = C.project_points(xyz_c, g["camera_K"], D_cam)
    pp = C.project_points(xyz_p, g["projector_K"], None)
    rc = C.undistort_points(pc, g["camera_K"], D_cam, R1, P1)   # camera pixel -> rectified
    rp = C.undistort_points(pp, g["projector_K"], None, R2, P2)
    assert np.abs(rc[:, 1] - rp[:, 1]).max() < 1e-3            # same row
    disp = rp[:, 0] - rc[:, 0]
    z_rect = (xyz_c @ R1.T)[:, 2]
    assert np.allclose(P2[0, 3] / disp, z_rect, rtol=1e-4)      # depth = P2[0,3] / disparity (disp_to_depth.py:58-61)
    # forward map (rectified -> source) composed with the inverse (source -> rectified) is the identity
    mx, my = C.init_undistort_rectify_map(g["camera_K"], D_cam, R1, P1, size)
    iy, ix = np.rint(rc[:, 1]).astype(int), np.rint(rc[:, 0]).astype(int)
    ok = (ix >= 0) & (ix < size[0]) & (iy >= 0) & (iy < size[1])
    assert ok.mean() > 0.9
    back = np.stack((mx[iy[ok], ix[ok]], my[iy[ok], ix[ok]]), -1)
    assert np.abs(back - pc[ok]).max() < 1.0                    # within the half-pixel rounding of (ix, iy)


def test_remap_and_i16_helpers():
    img = np.arange(12, dtype=np.float32).reshape(3, 4)
    mx = np.array([[-1.0, 0.4, 3.6, 9.0]], np.float32)
    my = np.array([[0.0, 1.0, 2.4, 1.0]], np.float32)
    assert np.array_equal(C.remap_nearest(img, mx, my, "replicate"), [[0, 4, 11, 7]])
    assert np.array_equal(C.remap_nearest(img, mx, my, "constant"), [[0, 4, 0, 0]])
    assert np.array_equal(C.mapf_to_i16(np.array([0.5, 1.5, -0.5, 2.4999], np.float32)), [0, 2, 0, 2])
