"""-m gpu: the EVT 3.0 decoder on the device (x_maps_amd/csrc/xmaps_evt3.hpp: the format's state machine as three scan kernels)
against the INDEPENDENT checker oracle/evt3_oracle.py (one word at a time, nothing shared with the product; itself pinned by the
hand-derived sequences of tests/test_evt3_oracle.py), word for word: those hand-derived sequences, random streams
with single and vector events, redundant / changing TIME_HIGH words, 24-bit wrap-arounds, skipped word types, words in front of
the first row / time word, any chunking (state carried on the device), and the decoder in front of the device ingest
(xm_ingest_push_evt3): the same frames as pushing the host-decoded packets."""
import numpy as np
import pytest

import evt3_oracle as EO
from x_maps_amd import XMapsEngine, evt3
from x_maps_amd import synthetic as S

pytestmark = pytest.mark.gpu


def W(typ, payload):
    return (typ << 12) | (payload & 0xfff)


def _same(a, b):
    return len(a) == len(b) and all(np.array_equal(a[k], b[k]) for k in ("x", "y", "p", "t"))


@pytest.fixture(scope="module")
def eng():
    with XMapsEngine(S.make_tables(S.C_TINY)) as e:
        yield e


def test_hand_derived_sequences(eng):
    """the word sequences whose events were worked out on paper (tests/test_evt3_oracle.py: HAND), whole and split anywhere"""
    from test_evt3_oracle import HAND, _rows
    for name, (words, want) in sorted(HAND.items()):
        w = np.array(words, dtype="<u2")
        with evt3.DeviceEvt3Decoder(eng, max_words=64) as dec:
            assert _rows(dec.decode(w)) == want, name
            for cut in range(len(words) + 1):
                dec.reset()
                assert _rows(dec.decode(w[:cut])) + _rows(dec.decode(w[cut:])) == want, (name, cut)


def _random_stream(seed, n_ev):
    rng = np.random.default_rng(700 + seed)
    cfg = S.RigConfig("evt3", 640, 480, 640, 480, n_ev)
    evs = S.make_events(cfg, frame=seed, n=n_ev, p_zero_fraction=0.3, t0=int(rng.choice([5_000_000, (1 << 24) - 6_000, 3 * (1 << 24) - 2_000])))
    # rows of simultaneous neighbours so that the encoder emits vectors
    k = n_ev // 40
    extra = np.zeros(6 * k, S.EVENT_CD_DTYPE)
    extra["t"] = np.repeat(evs["t"][:: max(n_ev // k, 1)][:k], 6)[: 6 * k]
    extra["y"] = np.repeat(rng.integers(0, 480, k), 6)
    extra["x"] = np.repeat(rng.integers(0, 600, k), 6) + np.tile(np.array([0, 1, 3, 4, 8, 11]), k)
    extra["p"] = np.repeat(rng.integers(0, 2, k), 6)
    allv = np.zeros(len(evs) + len(extra), S.EVENT_CD_DTYPE)
    allv[: len(evs)], allv[len(evs):] = evs, extra
    allv = allv[np.argsort(allv["t"], kind="stable")]
    words = evt3.encode_evt3(allv).astype(np.int64)
    # what a camera adds: redundant TIME_HIGH words, OTHERS / EXT_TRIGGER / CONTINUED words, stray empty vectors
    ins = np.sort(rng.integers(0, len(words), len(words) // 30))
    hi_at = np.maximum.accumulate(np.where((words >> 12) == 0x8, np.arange(len(words)), -1))
    filler = []
    for i in ins:
        r = rng.random()
        if r < 0.5 and hi_at[i] >= 0:
            filler.append(int(words[hi_at[i]]))  # the current TIME_HIGH once more
        elif r < 0.8:
            filler.append(int(W(int(rng.choice([0xE, 0xA, 0x7, 0xF])), int(rng.integers(0, 4096)))))
        else:
            filler.append(int(W(0x5, 0)))
    words = np.insert(words, ins, filler)
    return words.astype("<u2")


@pytest.mark.parametrize("seed", range(8))
def test_random_streams_whole_and_chunked(eng, seed):
    rng = np.random.default_rng(seed)
    words = _random_stream(seed, 30_000)
    ref = EO.decode(words)
    assert len(ref) > 30_000 and ((words >> 12) == 0x4).sum() > 100
    with evt3.DeviceEvt3Decoder(eng, max_words=len(words) + 8, max_events=len(ref) + 64) as dec:
        assert _same(dec.decode(words), ref)
        # chunks of any length, the state carried on the device; the checker in ONE go is the reference (streaming decoders)
        dec.reset()
        cuts = np.unique(np.concatenate(([0, len(words)], rng.integers(0, len(words), 12), [1, 2, 2049, 4096, 4097])))
        parts = [dec.decode(words[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
        cat = np.zeros(sum(len(p) for p in parts), S.EVENT_CD_DTYPE)
        o = 0
        for p in parts:
            cat[o:o + len(p)] = p
            o += len(p)
        assert _same(cat, ref)
    with evt3.DeviceEvt3Decoder(eng, max_words=1000) as dec:  # decode() splits by max_words
        assert _same(dec.decode(words), ref)


@pytest.mark.parametrize("seed", range(6))
def test_arbitrary_words(eng, seed):
    """any sequence of 16-bit words drives the state machine: uniformly random words (every type, vectors without bases, time
    fields jumping both ways, rows with the camera-id bit) decode the same on both sides, whole and in chunks"""
    rng = np.random.default_rng(900 + seed)
    n = int(rng.integers(1, 30_000))
    words = rng.integers(0, 65536, n).astype("<u2")
    if seed % 2:  # more of the words that carry state, fewer events
        sel = rng.random(n) < 0.5
        words[sel] = ((rng.choice([0x0, 0x3, 0x6, 0x8], int(sel.sum())) << 12) | rng.integers(0, 4096, int(sel.sum()))).astype("<u2")
    ref = EO.decode(words)
    with evt3.DeviceEvt3Decoder(eng, max_words=n + 8, max_events=len(ref) + 64) as dec:
        assert _same(dec.decode(words), ref)
        dec.reset()
        sm = EO.Evt3StateMachine()
        cuts = np.unique(np.concatenate(([0, n], rng.integers(0, n, 9))))
        for a, b in zip(cuts[:-1], cuts[1:]):
            assert _same(dec.decode(words[a:b]), sm.feed(words[a:b])), (a, b)


def test_limits_are_reported(eng):
    words = _random_stream(1, 4_000)
    with evt3.DeviceEvt3Decoder(eng, max_words=len(words), max_events=100) as dec:
        with pytest.raises(Exception):
            dec.decode_device(words)
    with evt3.DeviceEvt3Decoder(eng, max_words=100) as dec:
        with pytest.raises(Exception):
            dec.decode_device(words)


def test_in_front_of_the_device_ingest():
    """raw words -> xm_ingest_push_evt3 (count waited for / left on the device) == checker-decoded packets -> xm_ingest_push, packet
    by packet: same frames, same depth"""
    from x_maps_amd.ingest import DeviceIngest
    cfg = S.C_TINY
    tb = S.make_tables(cfg)
    fps = 60
    import test_gpu_ingest as TI
    stream = TI._tiny_stream(14, seed=7)  # frames without a 40 us pause inside, 3.6 ms between them
    chunks = [evt3.encode_evt3(pk) for pk in TI._packets(stream, int(1e6 / fps / 4)) if len(pk)]  # a quarter of a period per chunk
    sm = EO.Evt3StateMachine()
    with XMapsEngine(tb) as e1, XMapsEngine(tb) as e2, XMapsEngine(tb) as e3:
        ing1 = DeviceIngest(e1, fps, max_packet_events=8192, capacity_events=65536, result_ring=64)
        ing2 = DeviceIngest(e2, fps, max_packet_events=8192, capacity_events=65536, result_ring=64)
        ing3 = DeviceIngest(e3, fps, max_packet_events=8192, capacity_events=65536, result_ring=64)
        out1, out2, out3 = [], [], []
        pinned = []
        with evt3.DeviceEvt3Decoder(e1, max_words=max(len(c) for c in chunks)) as dec, \
                evt3.DeviceEvt3Decoder(e3, max_words=max(len(c) for c in chunks)) as dec3:
            for words in chunks:
                n_dev = dec.push(ing1, words)
                pkt = sm.feed(words)
                assert n_dev == len(pkt)
                ing2.push(pkt)
                # nothing waited for: the chunk's event count stays on the device (pinned words go through the launch thread)
                pw = e3.host_empty(words.shape, np.uint16)
                pw[:] = words
                pinned.append(pw)
                assert dec3.push(ing3, pw, pinned=True, count=False) is None
                out1 += ing1.poll()
                out2 += ing2.poll()
                out3 += ing3.poll()
            ing1.flush(); ing2.flush(); ing3.flush()
            out1 += ing1.poll(); out2 += ing2.poll(); out3 += ing3.poll()
        assert len(out1) == len(out2) == len(out3) >= 4
        for a, b, c in zip(out1, out2, out3):
            assert (a.n_events, a.t_first, a.t_last, a.n_inliers) == (b.n_events, b.t_first, b.t_last, b.n_inliers)
            assert (c.n_events, c.t_first, c.t_last, c.n_inliers, c.overflow) == (b.n_events, b.t_first, b.t_last, b.n_inliers, 0)
            assert np.array_equal(a.depth, b.depth) and np.array_equal(a.bgr, b.bgr)
            assert np.array_equal(c.depth, b.depth) and np.array_equal(c.bgr, b.bgr)
        ing1.close(); ing2.close(); ing3.close()


def test_words_and_records_take_turns_on_one_ingest():
    """a stream that arrives now as EVT 3.0 chunks, now as records (frames cut from chunks leave on the frame stream, frames cut
    from records on the out stream: the two take turns inside one ingest) == the same packets as records throughout"""
    from x_maps_amd.ingest import DeviceIngest
    import test_gpu_ingest as TI
    tb = S.make_tables(S.C_TINY)
    fps = 60
    stream = TI._tiny_stream(30, seed=11)
    packets = [pk for pk in TI._packets(stream, int(1e6 / fps / 3)) if len(pk)]
    with XMapsEngine(tb) as e1, XMapsEngine(tb) as e2:
        with DeviceIngest(e1, fps, max_packet_events=8192, capacity_events=65536, result_ring=64) as mixed, \
                DeviceIngest(e2, fps, max_packet_events=8192, capacity_events=65536, result_ring=64) as plain, \
                evt3.DeviceEvt3Decoder(e1, max_words=8 * 8192) as dec:
            got, want, keep = [], [], []
            for i, pk in enumerate(packets):
                if (i // 7) % 2 == 0:  # seven packets as words (count left on the device / waited for), seven as records, ...
                    w = evt3.encode_evt3(pk)
                    if i % 2:
                        pw = e1.host_empty(w.shape, np.uint16)
                        pw[:] = w
                        keep.append(pw)
                        dec.push(mixed, pw, pinned=True, count=False)
                    else:
                        assert dec.push(mixed, w) == len(pk)
                    # (each chunk was encoded on its own and starts with TIME_HIGH / TIME_LOW: the packets the decoder did not see
                    #  in between do not matter to it)
                else:
                    mixed.push(pk)
                plain.push(pk)
                if i % 5 == 0:
                    got += mixed.poll()
                    want += plain.poll()
            mixed.flush(); plain.flush()
            got += mixed.poll(); want += plain.poll()
    assert len(got) == len(want) >= 20 and not any(f.lost or f.overflow for f in got)
    for a, b in zip(got, want):
        assert (a.seq, a.n_events, a.t_first, a.t_last, a.n_inliers) == (b.seq, b.n_events, b.t_first, b.t_last, b.n_inliers)
        assert np.array_equal(a.depth, b.depth) and np.array_equal(a.bgr, b.bgr)


def test_a_raw_file_through_the_processor(tmp_path):
    """DepthReprojectionProcessor.process_evt3_words (device ingest: words decoded on the GPU; host ingest: on the host) shows the
    same frames as process_events on the decoded packets"""
    import test_gpu_ingest as TI
    from x_maps_amd.depth_reprojection_processor import DepthReprojectionProcessor, RuntimeParams
    cfg = S.C_TINY
    tb = S.make_tables(cfg)
    stream = TI._tiny_stream(14, seed=7)
    chunks = [evt3.encode_evt3(pk) for pk in TI._packets(stream, int(1e6 / 60 / 4)) if len(pk)]  # a quarter of a period per chunk
    path = tmp_path / "rec.raw"
    with open(path, "wb") as f:
        f.write(b"% evt 3.0\n% end\n")
        for c in chunks:
            f.write(c.tobytes())
    assert np.array_equal(np.concatenate(list(evt3.read_raw_words(str(path), chunk_words=777))), np.concatenate(chunks))
    host_dec = evt3.Evt3Decoder()
    packets = [host_dec.decode(c) for c in chunks]
    shown = {}
    for mode in ("words_device", "words_host", "records_device", "records_host"):
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
            if mode.startswith("records"):
                for ev in packets:
                    proc.process_events(ev)
            else:
                for w in chunks:
                    proc.process_evt3_words(w)
            proc.flush()
        shown[mode] = frames
    assert len(shown["records_host"]) >= 4
    for mode in ("words_device", "words_host", "records_device"):
        assert len(shown[mode]) == len(shown["records_host"]), mode
        assert all(np.array_equal(a, b) for a, b in zip(shown[mode], shown["records_host"])), mode


@pytest.mark.parametrize("fmt", [3, 2])
def test_wait_for_time_base_on_the_device(fmt):
    """the start-of-stream option of the device decoders (xm_evt3_wait_for_time_base) == the oracle with the same option: streams
    that start with events in front of their first EVT_TIME_HIGH, whole and in chunks (the flag lives in the decoder's device
    state), random word soups, through the ingest with the count left on the device"""
    import evt2_oracle
    import evt3_oracle
    from x_maps_amd import evt2
    rng = np.random.default_rng(70 + fmt)
    cfg = S.C_TINY
    ora = evt3_oracle if fmt == 3 else evt2_oracle
    with XMapsEngine(S.make_tables(cfg)) as eng:
        cls = evt3.DeviceEvt3Decoder if fmt == 3 else evt2.DeviceEvt2Decoder
        for trial in range(6):
            ev = S.make_events(cfg, frame=trial, n=4000)
            w = evt3.encode_evt3(ev) if fmt == 3 else evt2.encode_evt2(ev, time_high_every_us=16)
            hi = np.nonzero((w >> 12) == 0x8)[0] if fmt == 3 else np.nonzero((w >> 28) == 0x8)[0]
            w = np.delete(w, hi[:1 + trial % 3])            # the first one to three TIME_HIGH words are missing
            if trial >= 4:                                  # arbitrary words: every type, any order
                w = rng.integers(0, 1 << (16 if fmt == 3 else 32), 6000, dtype=np.uint64).astype(w.dtype)
            for wait in (False, True):
                want = ora.decode(w, wait)
                with cls(eng, max_words=1 << 14, wait_for_time_base=wait) as dec:
                    got = dec.decode(w)
                    assert len(got) == len(want) and all(np.array_equal(got[k], want[k]) for k in "xypt"), (trial, wait)
                    dec.reset()
                    cuts = [0] + sorted(rng.integers(1, len(w), 3).tolist()) + [len(w)]
                    parts = [dec.decode(w[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
                    got = np.concatenate(parts)
                    assert len(got) == len(want) and all(np.array_equal(got[k], want[k]) for k in "xypt"), (trial, wait, cuts)
