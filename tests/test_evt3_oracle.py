"""The independent EVT 3.0 checker (oracle/evt3_oracle.py: one word at a time) pinned by HAND-DERIVED word sequences -- the
expected events below were worked out on paper from the format description, not produced by any decoder -- and the product's
vectorised host decoder (x_maps_amd/evt3.py) held against it on random streams and on uniformly random words."""
import numpy as np
import pytest

import evt3_oracle as EO
from x_maps_amd import evt3


def W(typ, payload):
    return (typ << 12) | (payload & 0xfff)


def _rows(ev):
    return [(int(a), int(b), int(c), int(d)) for a, b, c, d in zip(ev["x"], ev["y"], ev["p"], ev["t"])]


P = 1 << 11  # the polarity bit of ADDR_X / VECT_BASE_X

HAND = {
    # TIME_HIGH 1 -> t = 4096; TIME_LOW 0x10 -> 4112; row 37; two singles; TIME_LOW 0x11 -> 4113; a vector base 200 (p = 1):
    # VECT_12 bits 0, 2, 11 -> 200, 202, 211, base 212; VECT_8 bits 0, 7 -> 212, 219, base 220; OTHERS / EXT_TRIGGER skipped
    "singles_and_vectors": (
        [W(0x8, 1), W(0x6, 0x10), W(0x0, 37), W(0x2, P | 100), W(0x2, 101), W(0x6, 0x11), W(0x3, P | 200),
         W(0x4, 0b100000000101), W(0x5, 0b10000001), W(0xE, 0x123), W(0xA, 1), W(0x0, 5), W(0x2, 7)],
        [(100, 37, 1, 4112), (101, 37, 0, 4112), (200, 37, 1, 4113), (202, 37, 1, 4113), (211, 37, 1, 4113),
         (212, 37, 1, 4113), (219, 37, 1, 4113), (7, 5, 0, 4113)]),
    # a stale TIME_LOW: high 5, low 4000 -> 24480; the SAME high again changes nothing; high 6 restarts the low field:
    # 6 * 4096 = 24576 (not 24576 + 4000); then TIME_LOW 3 -> 24579
    "stale_time_low": (
        [W(0x8, 5), W(0x6, 4000), W(0x0, 7), W(0x2, P | 10), W(0x8, 5), W(0x2, P | 11), W(0x8, 6), W(0x2, P | 12), W(0x6, 3),
         W(0x2, P | 13)],
        [(10, 7, 1, 24480), (11, 7, 1, 24480), (12, 7, 1, 24576), (13, 7, 1, 24579)]),
    # the 24-bit wrap: high 0xFFE / low 0xFFF -> 16773119 + ... = 0xFFEFFF; high 0xFFF -> 0xFFF000; high 0 (4095 below) is a
    # wrap: 2^24 + 0, low 2 -> 2^24 + 2; high 1 -> 2^24 + 4096
    "wrap_around": (
        [W(0x8, 0xffe), W(0x6, 0xfff), W(0x0, 1), W(0x2, 1), W(0x8, 0xfff), W(0x2, 2), W(0x8, 0), W(0x6, 2), W(0x2, 3), W(0x8, 1),
         W(0x2, 4)],
        [(1, 1, 0, 0xffefff), (2, 1, 0, 0xfff000), (3, 1, 0, (1 << 24) + 2), (4, 1, 0, (1 << 24) + 4096)]),
    # a small step back of the high field (jitter) is NOT a wrap: 0x20 -> 0x1f stays in loop 0
    "high_steps_back_a_little": (
        [W(0x8, 0x20), W(0x0, 2), W(0x2, 9), W(0x8, 0x1f), W(0x6, 5), W(0x2, 9)],
        [(9, 2, 0, 0x20000), (9, 2, 0, 0x1f005)]),
    # initial state: an event before any row / time word sits at (x, 0) at t = 0; vectors without a base start at column 0;
    # empty vectors only advance the base: VECT_12 0xfff -> 0..11 (base 12), VECT_8 0 (20), VECT_12 0 (32), base 9 (p = 0):
    # VECT_8 0xff -> 9..16 (base 17), VECT_8 0x81 -> 17, 24
    "initial_state_and_empty_vectors": (
        [W(0x2, 3), W(0x4, 0xfff), W(0x5, 0), W(0x4, 0), W(0x3, 9), W(0x5, 0xff), W(0x5, 0x81)],
        [(3, 0, 0, 0)] + [(i, 0, 0, 0) for i in range(12)] + [(9 + i, 0, 0, 0) for i in range(8)] + [(17, 0, 0, 0), (24, 0, 0, 0)]),
    # polarity flips: the base word's polarity holds for every vector behind it, a single's own bit for itself; bit 11 of
    # ADDR_Y (camera flag) is not part of the row; VECT_8 ignores payload bits 8..11
    "polarity_and_masks": (
        [W(0x0, P | 44), W(0x6, 1), W(0x3, 50), W(0x5, 0xf03), W(0x3, P | 50), W(0x5, 0x003), W(0x2, 1), W(0x2, P | 1)],
        [(50, 44, 0, 1), (51, 44, 0, 1), (50, 44, 1, 1), (51, 44, 1, 1), (1, 44, 0, 1), (1, 44, 1, 1)]),
    "nothing": ([], []),
    "only_skipped_words": ([W(0xA, 7), W(0xE, 1), W(0x7, 2), W(0xF, 3), W(0x1, 4), W(0x9, 5), W(0xB, 6), W(0xC, 7), W(0xD, 8)], []),
}


@pytest.mark.parametrize("name", sorted(HAND))
def test_oracle_reproduces_the_hand_derived_events(name):
    words, want = HAND[name]
    assert _rows(EO.decode(np.array(words, dtype="<u2"))) == want
    # ... and in any chunking (the state carries over)
    for cut in range(len(words) + 1):
        sm = EO.Evt3StateMachine()
        got = _rows(sm.feed(np.array(words[:cut], dtype="<u2"))) + _rows(sm.feed(np.array(words[cut:], dtype="<u2")))
        assert got == want, (name, cut)


@pytest.mark.parametrize("name", sorted(HAND))
def test_host_decoder_reproduces_the_hand_derived_events(name):
    words, want = HAND[name]
    assert _rows(evt3.decode_evt3(np.array(words, dtype="<u2"))) == want


def _same(a, b):
    return len(a) == len(b) and all(np.array_equal(a[k], b[k]) for k in ("x", "y", "p", "t"))


@pytest.mark.parametrize("seed", range(6))
def test_host_decoder_equals_the_oracle_on_arbitrary_words(seed):
    rng = np.random.default_rng(4100 + seed)
    n = int(rng.integers(1, 20_000))
    words = rng.integers(0, 65536, n).astype("<u2")
    if seed % 2:  # more state-carrying words
        sel = rng.random(n) < 0.5
        words[sel] = ((rng.choice([0x0, 0x3, 0x6, 0x8], int(sel.sum())) << 12) | rng.integers(0, 4096, int(sel.sum()))).astype("<u2")
    ref = EO.decode(words)
    assert _same(evt3.decode_evt3(words), ref)
    host, sm = evt3.Evt3Decoder(), EO.Evt3StateMachine()
    cuts = np.unique(np.concatenate(([0, n], rng.integers(0, n, 7))))
    for a, b in zip(cuts[:-1], cuts[1:]):
        assert _same(host.decode(words[a:b]), sm.feed(words[a:b])), (a, b)


@pytest.mark.parametrize("vectors", [True, False])
def test_encoders_round_trip_through_the_oracle(vectors):
    """what write_raw / the bench's word streams hold decodes back to the events that went in -- including two consecutive
    events whose low 12 bits are equal and whose high fields differ (the encoder must re-send TIME_LOW after TIME_HIGH)"""
    rng = np.random.default_rng(5)
    n = 5000
    ev = np.zeros(n, EO.EVENT_CD)
    ev["t"] = np.sort(rng.integers(0, 3 << 24, n))
    ev["t"][10:13] = [0x1005 + ev["t"][9] // 4096 * 4096 + 8192, 0x2005 + ev["t"][9] // 4096 * 4096 + 8192, 0x2006 + ev["t"][9] // 4096 * 4096 + 8192]
    ev["t"] = np.sort(ev["t"])
    ev["x"], ev["y"], ev["p"] = rng.integers(0, 640, n), rng.integers(0, 480, n), rng.integers(0, 2, n)
    words = evt3.encode_evt3(ev, use_vectors=True) if vectors else evt3.encode_evt3_singles(ev)
    assert _same(EO.decode(words), ev)
    tiny = np.zeros(3, EO.EVENT_CD)
    tiny["t"] = [0x1005, 0x2005, 0x2006]
    for enc in (evt3.encode_evt3, evt3.encode_evt3_singles):
        assert list(EO.decode(enc(tiny))["t"]) == [0x1005, 0x2005, 0x2006]
        assert list(evt3.decode_evt3(enc(tiny))["t"]) == [0x1005, 0x2005, 0x2006]


def test_wait_for_time_base_is_an_option_of_every_decoder():
    """Start-of-stream rule: events in front of the stream's first EVT_TIME_HIGH are emitted at time base 0 (default) or not at all
    (wait_for_time_base=True: a reader that waits for the first time base) -- oracle and host decoder, EVT 3.0 and 2.0, whole
    and chunked (the flag carries over: a chunk without any TIME_HIGH behind one that had it emits everything)."""
    import evt2_oracle
    import evt3_oracle
    from x_maps_amd import evt2
    pre3 = np.array([0x6000 | 77, 0x0000 | 9, 0x2000 | (1 << 11) | 5, 0x3000 | 40, 0x4000 | 0b101, 0x8000 | 2, 0x6000 | 3, 0x2000 | 6, 0x5000 | 0b11], "<u2")
    a, b = evt3_oracle.decode(pre3), evt3_oracle.decode(pre3, wait_for_time_base=True)
    assert len(a) == 1 + 2 + 1 + 2 and len(b) == 1 + 2 and list(a["t"][:3]) == [77, 77, 77] and list(b["t"]) == [(2 << 12) | 3] * 3
    for wait in (False, True):
        want = evt3_oracle.decode(pre3, wait)
        assert np.array_equal(evt3.decode_evt3(pre3, wait_for_time_base=wait), want)
        for cut in range(1, len(pre3)):
            d = evt3.Evt3Decoder(wait_for_time_base=wait)
            got = np.concatenate((d.decode(pre3[:cut]), d.decode(pre3[cut:])))
            assert np.array_equal(got, want), (wait, cut)
    pre2 = np.array([(1 << 28) | (5 << 22) | (7 << 11) | 3, (0 << 28) | (6 << 22) | (8 << 11) | 4, (8 << 28) | 100, (1 << 28) | (1 << 22) | (9 << 11) | 5,
                     (0xA << 28) | 1, (1 << 28) | (2 << 22) | (10 << 11) | 6], "<u4")
    for wait in (False, True):
        want = evt2_oracle.decode(pre2, wait)
        assert len(want) == (2 if wait else 4)
        assert np.array_equal(evt2.decode_evt2(pre2, wait_for_time_base=wait), want)
        for cut in range(1, len(pre2)):
            d = evt2.Evt2Decoder(wait_for_time_base=wait)
            got = np.concatenate((d.decode(pre2[:cut]), d.decode(pre2[cut:])))
            assert np.array_equal(got, want), (wait, cut)
