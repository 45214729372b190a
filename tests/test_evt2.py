"""CPU: the EVT 2.0 host decoder (x_maps_amd/evt2.py: forward fills over the whole buffer) against hand-derived word sequences
and against the independent word-by-word state machine (oracle/evt2_oracle.py), whole and in chunks; encoder round trip; RAW
files.  (Metavision's reader is closed: parity against it is unpinned)"""
import numpy as np
import pytest

import evt2_oracle as EO
from x_maps_amd import evt2, evt3
from x_maps_amd import synthetic as S


def cd(p, t6, x, y):
    return (p << 28) | (t6 << 22) | (x << 11) | y


def th(v):
    return (0x8 << 28) | v


def _same(a, b):
    return len(a) == len(b) and all(np.array_equal(a[k], b[k]) for k in ("x", "y", "p", "t"))


# (words, expected (x, y, p, t) tuples), derived by hand from the format description
HAND = [
    # no TIME_HIGH yet: the time base is 0
    ([cd(1, 5, 10, 20)], [(10, 20, 1, 5)]),
    # a time base of 3 -> t = 3 * 64 + low; both polarities; the largest coordinates and the largest low field
    ([th(3), cd(0, 0, 0, 0), cd(1, 63, 2047, 2047)], [(0, 0, 0, 192), (2047, 2047, 1, 255)]),
    # the base changes between events; a repeated TIME_HIGH changes nothing; triggers / vendor / continued words are skipped
    ([th(1), cd(1, 1, 1, 1), th(1), (0xA << 28) | 0x123, cd(1, 2, 1, 1), th(2), (0xE << 28) | 7, (0xF << 28) | 9, cd(0, 0, 5, 6)],
     [(1, 1, 1, 65), (1, 1, 1, 66), (5, 6, 0, 128)]),
    # the 28-bit field wraps: 0x0fffffff -> 0 is one loop = 2^34 us more
    ([th(0x0FFFFFFF), cd(1, 63, 1, 2), th(0), cd(1, 0, 3, 4)], [(1, 2, 1, (0x0FFFFFFF << 6) | 63), (3, 4, 1, 1 << 34)]),
    # a small step BACK is not a loop (out-of-order TIME_HIGH words of a camera that interleaves two sources)
    ([th(100), cd(1, 0, 1, 1), th(99), cd(1, 0, 2, 2)], [(1, 1, 1, 6400), (2, 2, 1, 6336)]),
    # unknown types carry nothing
    ([(0x2 << 28) | 5, (0x7 << 28) | 5, th(1), (0x9 << 28) | 1, cd(1, 3, 4, 5)], [(4, 5, 1, 67)]),
]


@pytest.mark.parametrize("case", range(len(HAND)))
def test_hand_derived_sequences(case):
    words, want = HAND[case]
    w = np.array(words, dtype="<u4")
    for dec in (evt2.decode_evt2(w), EO.decode(w)):
        got = [(int(e["x"]), int(e["y"]), int(e["p"]), int(e["t"])) for e in dec]
        assert got == want


@pytest.mark.parametrize("seed", range(6))
def test_random_streams_whole_and_chunked(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(100, 6000))
    ev = np.zeros(n, S.EVENT_CD_DTYPE)
    ev["t"] = np.sort(rng.integers(0, 1 << 22, n)) + int(rng.integers(0, 1 << 33))
    ev["x"], ev["y"], ev["p"] = rng.integers(0, 1280, n), rng.integers(0, 720, n), rng.integers(0, 2, n)
    w = evt2.encode_evt2(ev, time_high_every_us=int(rng.choice([0, 16, 1000])))
    assert _same(evt2.decode_evt2(w), ev) and _same(EO.decode(w), ev)
    dec, sm, parts, ref = evt2.Evt2Decoder(), EO.Evt2StateMachine(), [], []
    cuts = np.sort(rng.integers(0, len(w) + 1, 7))
    for a, b in zip(np.concatenate(([0], cuts)), np.concatenate((cuts, [len(w)]))):
        parts.append(dec.decode(w[a:b]))
        ref.append(sm.feed(w[a:b]))
        assert _same(parts[-1], ref[-1])
    assert sum(len(p) for p in parts) == n


@pytest.mark.parametrize("seed", range(6))
def test_arbitrary_words(seed):
    """any 32-bit words at all (every type, loops, nothing sorted): the two decoders agree word for word, chunked or not"""
    rng = np.random.default_rng(100 + seed)
    n = 4000
    w = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype("<u4")
    kinds = rng.choice([0x0, 0x1, 0x8, 0xA, 0xE, 0xF, 0x3], n, p=[0.3, 0.3, 0.25, 0.05, 0.04, 0.03, 0.03]).astype(np.uint32)
    w = (w & np.uint32(0x0FFFFFFF)) | (kinds << np.uint32(28))
    assert _same(evt2.decode_evt2(w), EO.decode(w))
    dec, sm = evt2.Evt2Decoder(), EO.Evt2StateMachine()
    for a in range(0, n, 333):
        assert _same(dec.decode(w[a:a + 333]), sm.feed(w[a:a + 333]))
    assert (dec.t_high, dec.t_loops) == (sm.time_high, sm.loops)


def test_empty_and_eventless_chunks():
    dec = evt2.Evt2Decoder()
    assert len(dec.decode(np.zeros(0, "<u4"))) == 0
    assert len(dec.decode(np.array([th(7), (0xE << 28)], "<u4"))) == 0 and dec.t_high == 7
    assert int(dec.decode(np.array([cd(1, 1, 2, 3)], "<u4"))["t"][0]) == 7 * 64 + 1  # the base survives chunks without events


def test_raw_file_round_trip_and_the_wrong_format(tmp_path):
    ev = S.make_events(S.C_TINY, frame=2, n=3000, p_zero_fraction=0.4)
    p = tmp_path / "rec2.raw"
    evt2.write_raw(str(p), ev, width=S.C_TINY.cam_w, height=S.C_TINY.cam_h)
    got = list(evt2.read_raw(str(p), chunk_words=1000))
    cat = np.zeros(sum(len(g) for g in got), S.EVENT_CD_DTYPE)
    o = 0
    for g in got:
        cat[o:o + len(g)] = g
        o += len(g)
    assert _same(cat, ev) and len(got) > 1
    assert sum(len(w) for w in evt2.read_raw_words(str(p), chunk_words=512)) == len(evt2.encode_evt2(ev))
    with pytest.raises(ValueError, match="EVT 3.0"):  # each reader refuses the other encoding and says which one the file holds
        list(evt3.read_raw(str(p)))
    p3 = tmp_path / "rec3.raw"
    evt3.write_raw(str(p3), ev)
    with pytest.raises(ValueError, match="EVT 2.0"):
        list(evt2.read_raw(str(p3)))
