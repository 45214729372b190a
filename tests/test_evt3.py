"""CPU: the EVT 3.0 decoder (x_maps_amd/evt3.py) on hand-built word sequences, a round trip through the encoder, chunked
decoding (state carried across chunks) and a 24-bit time wrap."""
import numpy as np
import pytest

from x_maps_amd import evt3
from x_maps_amd import synthetic as S


def W(typ, payload):
    return (typ << 12) | (payload & 0xfff)


def test_hand_built_words():
    words = np.array([
        W(0x8, 0x001), W(0x6, 0x010),             # t = 0x001010 = 4112
        W(0x0, 37),                               # y = 37
        W(0x2, (1 << 11) | 100),                  # single: x 100, p 1
        W(0x2, 101),                              # single: x 101, p 0
        W(0x6, 0x011),                            # t = 4113
        W(0x3, (1 << 11) | 200), W(0x4, 0b100000000101), W(0x5, 0b10000001),  # vectors: base 200 p 1; 12-bit; then base 212, 8-bit
        W(0xE, 0x123), W(0xA, 0x001),             # OTHERS / EXT_TRIGGER: skipped
        W(0x0, 5), W(0x2, 7),                     # y = 5, x = 7, p 0
    ], dtype="<u2")
    ev = evt3.decode_evt3(words)
    assert list(ev["x"]) == [100, 101, 200, 202, 211, 212, 219, 7]
    assert list(ev["y"]) == [37, 37, 37, 37, 37, 37, 37, 5]
    assert list(ev["p"]) == [1, 0, 1, 1, 1, 1, 1, 0]
    assert list(ev["t"]) == [4112, 4112, 4113, 4113, 4113, 4113, 4113, 4113]


def test_round_trip_and_chunked_decoding(tmp_path):
    rng = np.random.default_rng(4)
    evs = S.make_events(S.C_TINY, frame=2, n=20_000, p_zero_fraction=0.3)
    # rows of simultaneous neighbours so that the encoder emits vectors
    extra = np.zeros(600, S.EVENT_CD_DTYPE)
    extra["t"] = np.repeat(evs["t"][::100][:100], 6)
    extra["y"] = np.repeat(rng.integers(0, 48, 100), 6)
    extra["x"] = (np.repeat(rng.integers(0, 50, 100), 6) + np.tile(np.array([0, 1, 3, 4, 8, 11]), 100))
    extra["p"] = 1
    allv = np.concatenate((evs, extra))
    allv = allv[np.argsort(allv["t"], kind="stable")]
    words = evt3.encode_evt3(allv)
    assert ((words >> 12) == 0x4).sum() >= 50  # vectors really used
    dec = evt3.decode_evt3(words)
    for k in ("x", "y", "p", "t"):
        assert np.array_equal(dec[k], allv[k]), k
    # chunked: any split point, state carried over
    d = evt3.Evt3Decoder()
    parts = [d.decode(words[a:b]) for a, b in ((0, 7), (7, 5001), (5001, 5002), (5002, len(words)))]
    cat = np.concatenate(parts)
    assert np.array_equal(cat, dec)
    # file with header
    path = tmp_path / "rec.raw"
    evt3.write_raw(str(path), allv, 64, 48)
    got = np.concatenate(list(evt3.read_raw(str(path), chunk_words=3000)))
    assert np.array_equal(got, dec)
    fields, off = evt3.split_raw_header(path.read_bytes())
    assert fields["evt"] == "3.0" and off > 0


def test_time_wraps_after_24_bits():
    ev = np.zeros(4, S.EVENT_CD_DTYPE)
    ev["t"] = [(1 << 24) - 3, (1 << 24) - 1, (1 << 24) + 2, (1 << 24) + 5000]
    ev["x"], ev["y"], ev["p"] = [1, 2, 3, 4], 9, 1
    words = evt3.encode_evt3(ev)  # the encoder emits only the low 24 bits
    dec = evt3.decode_evt3(words)
    assert list(dec["t"] - dec["t"][0]) == [0, 2, 5, 5003]
    assert dec["t"][2] > dec["t"][1]  # monotone across the wrap


def test_time_high_change_restarts_the_low_field():
    """A TIME_HIGH word that changes the high field is followed by its TIME_LOW word only later: events in between carry low = 0,
    not the stale low of the previous period (stamps would jump up to 4095 us ahead and then run backwards)."""
    from x_maps_amd import evt3 as E
    TH, TL, Y, X = E.T_TIME_HIGH << 12, E.T_TIME_LOW << 12, E.T_ADDR_Y << 12, E.T_ADDR_X << 12
    words = np.array([TH | 5, TL | 4000, Y | 7, X | (1 << 11) | 10,   # t = 5 << 12 | 4000
                      TH | 5, X | (1 << 11) | 11,                      # redundant TIME_HIGH: nothing changes
                      TH | 6, X | (1 << 11) | 12,                      # high changed, no TIME_LOW yet: low = 0
                      TL | 3, X | (1 << 11) | 13], dtype="<u2")
    ev = E.decode_evt3(words)
    assert list(ev["x"]) == [10, 11, 12, 13]
    assert list(ev["t"]) == [(5 << 12) | 4000, (5 << 12) | 4000, 6 << 12, (6 << 12) | 3]
    assert np.all(np.diff(ev["t"]) >= 0)
    # the same stream split between the TIME_HIGH change and the event behind it
    dec = E.Evt3Decoder()
    a, b = dec.decode(words[:7]), dec.decode(words[7:])
    assert list(np.concatenate((a["t"], b["t"]))) == list(ev["t"])


def test_read_raw_checks_the_format_token(tmp_path):
    from x_maps_amd import evt3 as E
    p = tmp_path / "x.raw"
    p.write_bytes(b"% format EVT2;height=320;width=320\n% end\n" + b"\x00" * 8)
    with pytest.raises(ValueError):
        list(E.read_raw(str(p)))
    p.write_bytes(b"% format EVT3;height=320;width=320\n% end\n")
    assert list(E.read_raw(str(p))) == []


def test_vectorised_encoder_round_trips():
    evs = S.make_events(S.C_TINY, frame=4, n=30_000, p_zero_fraction=0.4, t0=(1 << 24) - 5_000)  # across a 24-bit wrap
    words = evt3.encode_evt3_singles(evs)
    assert np.array_equal(words, evt3.encode_evt3(evs, use_vectors=False))  # the same words as the file writer's loop
    dec = evt3.decode_evt3(words)
    assert np.array_equal(dec["x"], evs["x"]) and np.array_equal(dec["y"], evs["y"]) and np.array_equal(dec["p"], evs["p"])
    assert np.array_equal(dec["t"] - dec["t"][0], evs["t"] - evs["t"][0])  # (the format carries 24 bits of time)
    assert len(evt3.encode_evt3_singles(evs[:0])) == 0
