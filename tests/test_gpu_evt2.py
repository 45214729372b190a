"""-m gpu: the device EVT 2.0 decoder (csrc/xmaps_evt2.hpp: the word state machine as three scan launches) against the
independent word-by-word checker (oracle/evt2_oracle.py) and the hand-derived sequences of tests/test_evt2.py -- whole, in
chunks (state carried on the device), on arbitrary words; in front of the device ingest (== the same packets as records); a RAW
file through DepthReprojectionProcessor.process_evt2_words."""
import numpy as np
import pytest

import evt2_oracle as EO
from x_maps_amd import XMapsEngine, evt2, evt3
from x_maps_amd import synthetic as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    with XMapsEngine(S.make_tables(S.C_TINY)) as e:
        yield e


def _same(a, b):
    return len(a) == len(b) and all(np.array_equal(a[k], b[k]) for k in ("x", "y", "p", "t"))


def test_hand_derived_sequences(eng):
    from test_evt2 import HAND
    for words, want in HAND:
        with evt2.DeviceEvt2Decoder(eng, max_words=64) as dec:
            got = dec.decode(np.array(words, dtype="<u4"))
        assert [(int(e["x"]), int(e["y"]), int(e["p"]), int(e["t"])) for e in got] == want


@pytest.mark.parametrize("seed", range(5))
def test_random_streams_whole_and_chunked(eng, seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(2000, 60_000))
    ev = np.zeros(n, S.EVENT_CD_DTYPE)
    ev["t"] = np.sort(rng.integers(0, 1 << 23, n)) + int(rng.integers(0, 1 << 33))
    ev["x"], ev["y"], ev["p"] = rng.integers(0, 1280, n), rng.integers(0, 720, n), rng.integers(0, 2, n)
    words = evt2.encode_evt2(ev, time_high_every_us=int(rng.choice([0, 16, 1000])))
    ref = EO.decode(words)
    assert _same(ref, ev)
    with evt2.DeviceEvt2Decoder(eng, max_words=len(words) + 8) as dec:
        assert _same(dec.decode(words), ref)
        dec.reset()
        sm = EO.Evt2StateMachine()
        cuts = np.sort(rng.integers(0, len(words) + 1, 9))
        for a, b in zip(np.concatenate(([0], cuts)), np.concatenate((cuts, [len(words)]))):
            assert _same(dec.decode(words[a:b]), sm.feed(words[a:b])), (a, b)
    with evt2.DeviceEvt2Decoder(eng, max_words=3000) as dec:  # decode() splits by max_words: several blocks + chunks
        assert _same(dec.decode(words), ref)


@pytest.mark.parametrize("seed", range(5))
def test_arbitrary_words(eng, seed):
    rng = np.random.default_rng(200 + seed)
    n = 50_000
    w = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype("<u4")
    kinds = rng.choice([0x0, 0x1, 0x8, 0xA, 0xE, 0xF, 0x3], n, p=[0.3, 0.3, 0.25, 0.05, 0.04, 0.03, 0.03]).astype(np.uint32)
    w = (w & np.uint32(0x0FFFFFFF)) | (kinds << np.uint32(28))  # (loops: the TIME_HIGH values jump all over their 28 bits)
    ref = EO.decode(w)
    with evt2.DeviceEvt2Decoder(eng, max_words=n + 8) as dec:
        assert _same(dec.decode(w), ref)
        dec.reset()
        sm = EO.Evt2StateMachine()
        for a in range(0, n, 7777):
            assert _same(dec.decode(w[a:a + 7777]), sm.feed(w[a:a + 7777]))


def test_limits_and_the_wrong_encoding_are_reported(eng):
    words = evt2.encode_evt2(S.make_events(S.C_TINY, frame=1, n=500))
    with evt2.DeviceEvt2Decoder(eng, max_words=len(words), max_events=300) as dec:
        with pytest.raises(Exception, match="decodes to 500 events"):
            dec.decode_device(words)
        # (the state did not advance: the same chunk in two halves gives the whole stream)
        h = len(words) // 2
        assert _same(_cat(dec.decode(words[:h]), dec.decode(words[h:])), EO.decode(words))
    with evt2.DeviceEvt2Decoder(eng, max_words=100) as dec:
        with pytest.raises(Exception, match="max_words"):
            dec.decode_device(words)
    with evt3.DeviceEvt3Decoder(eng, max_words=1000) as d3:  # an EVT 3.0 decoder refuses 32-bit words and vice versa
        import ctypes as C
        n = C.c_size_t(0)
        rc = d3._lib.xm_evt2_decode(d3._d, C.c_void_p(words.ctypes.data), 10, None, C.byref(n))
        assert rc != 0 and "EVT 3.0" in d3._N.last_error()


def _cat(*parts):
    out = np.zeros(sum(len(p) for p in parts), S.EVENT_CD_DTYPE)
    o = 0
    for p in parts:
        out[o:o + len(p)] = p
        o += len(p)
    return out


def test_in_front_of_the_device_ingest():
    """raw words -> xm_ingest_push_evt2 (count waited for / left on the device, pageable / pinned) == the same packets as records"""
    from x_maps_amd.ingest import DeviceIngest
    import test_gpu_ingest as TI
    tb = S.make_tables(S.C_TINY)
    fps = 60
    stream = TI._tiny_stream(16, seed=5)
    packets = [pk for pk in TI._packets(stream, int(1e6 / fps / 4)) if len(pk)]
    with XMapsEngine(tb) as e1, XMapsEngine(tb) as e2:
        with DeviceIngest(e1, fps, max_packet_events=8192, capacity_events=65536, result_ring=64) as iw, \
                DeviceIngest(e2, fps, max_packet_events=8192, capacity_events=65536, result_ring=64) as ir, \
                evt2.DeviceEvt2Decoder(e1, max_words=2 * 8192) as dec:
            got, want, keep = [], [], []
            for i, pk in enumerate(packets):
                w = evt2.encode_evt2(pk, time_high_every_us=16)
                if i % 3 == 0:
                    assert dec.push(iw, w) == len(pk)
                elif i % 3 == 1:
                    assert dec.push(iw, w, count=False) is None
                else:
                    pw = e1.host_empty(w.shape, np.uint32)
                    pw[:] = w
                    keep.append(pw)
                    assert dec.push(iw, pw, pinned=True, count=False) is None
                ir.push(pk)
                got += iw.poll()
                want += ir.poll()
            iw.flush(); ir.flush()
            got += iw.poll(); want += ir.poll()
    assert len(got) == len(want) >= 6 and not any(f.lost or f.overflow for f in got)
    for a, b in zip(got, want):
        assert (a.n_events, a.t_first, a.t_last, a.n_inliers) == (b.n_events, b.t_first, b.t_last, b.n_inliers)
        assert np.array_equal(a.depth, b.depth) and np.array_equal(a.bgr, b.bgr)


def test_a_raw_file_through_the_processor(tmp_path):
    """DepthReprojectionProcessor.process_evt2_words (device ingest: decoded on the GPU; host trigger finder: on the host) shows
    the same frames as process_events on the decoded packets"""
    import test_gpu_ingest as TI
    from x_maps_amd.depth_reprojection_processor import DepthReprojectionProcessor, RuntimeParams
    cfg = S.C_TINY
    tb = S.make_tables(cfg)
    stream = TI._tiny_stream(12, seed=9)
    path = tmp_path / "rec2.raw"
    evt2.write_raw(str(path), stream, width=cfg.cam_w, height=cfg.cam_h)
    chunks = list(evt2.read_raw_words(str(path), chunk_words=6000))
    host_dec = evt2.Evt2Decoder()
    packets = [host_dec.decode(c) for c in chunks]
    shown = {}
    for mode in ("words_device", "words_host", "records_host"):
        frames = []

        class Window:
            def should_close(self):
                return False

            def show_async(self, img, acc=frames):
                acc.append(np.array(img))

        params = RuntimeParams(camera_width=cfg.cam_w, camera_height=cfg.cam_h, projector_width=cfg.proj_w,
                               projector_height=cfg.proj_h, projector_fps=60, z_near=0.1, z_far=1.2, calib=None,
                               projector_time_map=None, no_frame_dropping=True, camera_perspective=False, tables=tb,
                               device_ingest=mode.endswith("device"))
        with DepthReprojectionProcessor(params, window=Window()) as proc:
            for w, ev in zip(chunks, packets):
                if mode.startswith("words"):
                    proc.process_evt2_words(w)
                else:
                    proc.process_events(ev)
            proc.flush()
        shown[mode] = frames
    assert len(shown["records_host"]) >= 4
    for mode in ("words_device", "words_host"):
        assert len(shown[mode]) == len(shown["records_host"]), mode
        assert all(np.array_equal(a, b) for a, b in zip(shown[mode], shown["records_host"])), mode
