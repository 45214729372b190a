"""The owner-tile analysis of xm_create (csrc/xmaps_hip.hip: own_plan) is host code: xm_own_plan_info runs it without a device.
Which rigs qualify, and with what geometry -- including the ESL-like rig built from the reference's calibration numbers
(its X-map comes from the oracle's CPU builder here; on the GPU box the product's own kernel builds it)."""
import ctypes as C

import numpy as np
import pytest

import xmaps_oracle as O
from x_maps_amd import _native as N
from x_maps_amd import rig
from x_maps_amd import synthetic as S


def _plan(tb, grouped=True):
    """grouped: ownership of a cell per 8-row group where the rig allows it (the default), else / False: per row"""
    lib = N.load_library()
    N.debug_option("XM_OWN_GROUPED", None if grouped else "0")
    mapx = np.ascontiguousarray(tb["cam_mapx_i16"], np.int16)
    mapy = np.ascontiguousarray(tb["cam_mapy_i16"], np.int16)
    xmap = np.ascontiguousarray(tb["proj_x_map"], np.int16)
    cfg = N.xm_config()
    cfg.struct_size = C.sizeof(N.xm_config)
    cfg.cam_height, cfg.cam_width = mapx.shape
    cfg.rect_width, cfg.rect_height = int(tb["rect_w"]), int(tb["rect_h"])
    cfg.xmap_height, cfg.xmap_width = xmap.shape
    cfg.x_offset = int(tb.get("x_offset", 4242))
    cfg.cam_mapx_i16, cfg.cam_mapy_i16, cfg.proj_x_map = mapx.ctypes.data, mapy.ctypes.data, xmap.ctypes.data
    a = (C.c_int32 * 12)()
    try:
        N.check(lib.xm_own_plan_info(C.byref(cfg), a))
    finally:
        N.debug_option("XM_OWN_GROUPED", None)
    keys = ("mode", "w", "halo", "nxs_max", "shear_m", "shear_extra", "r_lo", "rows", "extras", "extras_max_per_tile", "delta_max",
            "lds_bytes")
    return dict(zip(keys, a))


@pytest.mark.parametrize("cpc,slant", [(3.3, -0.4), (2.0, 0.35), (4.6, -0.7), (1.4, 0.0), (7.5, -0.2)])
def test_shared_cell_rigs_qualify(cpc, slant):
    tb = S.make_tables_shared_cells(S.C_SHARED, cols_per_cell=cpc, slant=slant)
    p = _plan(tb, grouped=False)
    assert p["mode"] == 2 and p["halo"] == p["delta_max"] and p["w"] == 8 and 1 <= p["nxs_max"] <= 16, p  # (halo = the largest column distance inside a cell)
    assert p["delta_max"] == int(np.ceil(cpc)) - 1 or p["delta_max"] == int(np.ceil(cpc)), p
    # ownership per 8-row group where the column distance inside a cell of the GROUP still fits (the X-map's slant over 8 rows adds to it)
    pg = _plan(tb)
    assert pg["mode"] == 2 and pg["halo"] == pg["delta_max"] and p["delta_max"] <= pg["delta_max"] <= 7, pg
    assert (p["shear_m"] == 0) == (slant == 0.0), p
    # the frame's shear undoes the slant: (rows / 8) groups x m / 4096 columns
    assert abs(p["shear_extra"] - abs(slant) * S.C_SHARED.rect_h) <= 3, p
    assert p["lds_bytes"] < 16 * 1024


def test_too_many_columns_per_cell_or_too_wide_tiles_do_not_qualify():
    assert _plan(S.make_tables_shared_cells(S.C_SHARED, cols_per_cell=9.5))["mode"] == 0  # delta > 7
    # an injective X-map (2.1 cells per time column): the plain column tiles' business
    assert _plan(S.make_tables(S.C_1M))["mode"] == 0


def test_esl_like_rig_qualifies():
    """The reference's calibration numbers through the (pinned) rectification: 1080 time columns on ~780 frame columns, slant
    -0.40 columns per row: owner tiles of 8 columns + a halo of delta_max = 2, a band of <= 12 frame columns, extras in the first and the last tiles
    (where the rectified time map replicates its border or leaves the frame)."""
    cp, tb, evs, _ = rig.make_esl_like(row_stride=13, x_map_fn=lambda tm, *a: O.compute_x_map_from_time_map(np.asarray(tm, np.float32), *a))
    p = _plan(tb, grouped=False)
    assert p["mode"] == 2 and p["w"] == 8 and p["halo"] == p["delta_max"] and 1 <= p["delta_max"] <= 3, p
    pg = _plan(tb)  # per 8-row group: the slant of -0.40 columns per row over 8 rows x 1.4 time columns per cell on top
    assert pg["mode"] == 2 and pg["halo"] == pg["delta_max"] and 4 <= pg["delta_max"] <= 7, pg
    assert pg["w"] in (16, 20) and pg["nxs_max"] <= 24 and pg["lds_bytes"] <= 60 * 1024, pg  # wide tiles: the halo costs 1.35-1.44 x event reads
    assert p["nxs_max"] <= 12 and p["extras"] < 4000 and p["extras_max_per_tile"] <= 2048, p
    assert p["lds_bytes"] <= 60 * 1024, p  # two tiles per CU at least
    assert p["shear_m"] > 0 and p["shear_extra"] > 100, p  # the X-map is strongly slanted
    lo, hi = max(0, int(tb["cam_mapy_i16"].min())), min(int(tb["cam_mapy_i16"].max()), tb["rect_h"] - 2)
    assert p["r_lo"] % 8 == 0 and p["r_lo"] <= lo and p["r_lo"] + p["rows"] - 1 == hi
