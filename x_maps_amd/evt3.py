"""Prophesee EVT 3.0 RAW decoder ("next" row N4): the reader in front of the pipe, without the Metavision SDK.

The reference reads its recordings through Metavision's closed readers (python/bias_events_iterator.py:53-96:
RawReaderBase(input_filename, delta_t).load_delta_t(-1) yields EventCD packets).  EVT 3.0 itself is a public, documented
format: an ASCII header (lines starting with '%'), then little-endian 16-bit words whose top 4 bits are the type:

    0x0 EVT_ADDR_Y    y[10:0]                          (bit 11: camera id)      sets the current row
    0x2 EVT_ADDR_X    x[10:0], polarity bit 11                                  ONE event at (x, current y, current t)
    0x3 VECT_BASE_X   x[10:0], polarity bit 11                                  base column + polarity of the vectors below
    0x4 VECT_12       12 valid bits                                             events at base + i for every set bit; base += 12
    0x5 VECT_8        8 valid bits                                              same with 8; base += 8
    0x6 EVT_TIME_LOW  t[11:0]          0x8 EVT_TIME_HIGH  t[23:12]              current time (us), 24 bits, wraps every 16.8 s
    0xA EXT_TRIGGER, 0xE OTHERS, 0x7 / 0xF CONTINUED                            skipped here

The decoder is a state machine over the words; `decode_evt3` evaluates it for a whole buffer at once with forward fills
(np.maximum.accumulate of "index of the last word of type T") -- the same shape a device kernel would have (scan + gather).
PARITY: unpinned against Metavision (closed, and no recording ships with the reference); pinned by a round trip through the
encoder below and by hand-built word sequences (tests/test_evt3.py).
"""
from __future__ import annotations

import numpy as np

from .synthetic import EVENT_CD_DTYPE

T_ADDR_Y, T_ADDR_X, T_VECT_BASE_X, T_VECT_12, T_VECT_8, T_TIME_LOW, T_TIME_HIGH = 0x0, 0x2, 0x3, 0x4, 0x5, 0x6, 0x8


def split_raw_header(blob: bytes) -> tuple[dict, int]:
    """ASCII header of a .raw file: lines '% key value'; returns (fields, offset of the first data byte)."""
    off, fields = 0, {}
    while off < len(blob) and blob[off:off + 1] == b"%":
        end = blob.find(b"\n", off)
        if end < 0:
            break
        line = blob[off + 1:end].decode("ascii", "replace").strip()
        off = end + 1
        if line == "end":
            break
        k, _, v = line.partition(" ")
        fields[k] = v
    return fields, off


def _ffill_index(mask: np.ndarray) -> np.ndarray:
    """index of the last True at or before each position (-1 if none)"""
    idx = np.where(mask, np.arange(len(mask)), -1)
    return np.maximum.accumulate(idx)


class Evt3Decoder:
    """Streaming decoder: feed chunks of words, get EventCD arrays; state (row, base column, time, overflow count) carries over."""

    def __init__(self, wait_for_time_base: bool = False):
        self.y = 0
        self.base_x = 0
        self.base_p = 0
        self.t_low = 0
        self.t_high = 0
        self.t_loops = 0  # number of 24-bit wrap-arounds seen
        # start-of-stream rule: True = events in front of the stream's first EVT_TIME_HIGH word are not emitted (a reader that waits
        # for the first time base); False (default) = they carry the initial time base 0.  Unpinned against Metavision.
        self.wait_for_time_base = bool(wait_for_time_base)
        self.have_time = False  # an EVT_TIME_HIGH word has been read

    def decode(self, words: np.ndarray) -> np.ndarray:
        w = np.ascontiguousarray(words, dtype="<u2").astype(np.int64)
        n = len(w)
        if n == 0:
            return np.zeros(0, EVENT_CD_DTYPE)
        typ = w >> 12
        pos = np.arange(n)
        # ---- time: last TIME_HIGH / TIME_LOW before each word; 24-bit wrap-arounds counted on the TIME_HIGH sequence ----
        ih, il = _ffill_index(typ == T_TIME_HIGH), _ffill_index(typ == T_TIME_LOW)
        hi_words = np.nonzero(typ == T_TIME_HIGH)[0]
        th_seq = w[hi_words] & 0xfff
        prev = np.concatenate(([self.t_high], th_seq[:-1])) if len(th_seq) else th_seq
        # a wrap: the 12-bit high field falls back by (much) more than jitter would explain
        wraps = np.cumsum((prev - th_seq) > 0x800) if len(th_seq) else np.zeros(0, np.int64)
        loops_at = np.full(n, self.t_loops, np.int64)
        if len(hi_words):
            loops_at = np.where(ih >= 0, self.t_loops + wraps[np.searchsorted(hi_words, np.maximum(ih, 0))], self.t_loops)
        t_high = np.where(ih >= 0, w[np.maximum(ih, 0)] & 0xfff, self.t_high)
        t_low = np.where(il >= 0, w[np.maximum(il, 0)] & 0xfff, self.t_low)
        if len(hi_words):
            # a TIME_HIGH word that CHANGES the high field restarts the low field at 0 until the next TIME_LOW word (the stale
            # low would put the stamp up to 4095 us too late, and stamps must not run backwards when the TIME_LOW arrives);
            # the redundant TIME_HIGH words that repeat the current value change nothing
            changed = np.concatenate(([th_seq[0] != self.t_high], th_seq[1:] != th_seq[:-1]))
            last_change = _ffill_index(np.isin(pos, hi_words[changed]))
            t_low = np.where(last_change > il, 0, t_low)
        t = (loops_at << 24) | (t_high << 12) | t_low
        # ---- row: last ADDR_Y ----
        iy = _ffill_index(typ == T_ADDR_Y)
        y = np.where(iy >= 0, w[np.maximum(iy, 0)] & 0x7ff, self.y)
        # ---- single events ----
        sx = np.nonzero(typ == T_ADDR_X)[0]
        # ---- vector events: base column of each VECT word = last VECT_BASE_X + what earlier VECT words since it consumed ----
        adv = np.where(typ == T_VECT_12, 12, np.where(typ == T_VECT_8, 8, 0))
        cum = np.cumsum(adv)
        ib = _ffill_index(typ == T_VECT_BASE_X)
        base = np.where(ib >= 0, w[np.maximum(ib, 0)] & 0x7ff, self.base_x)
        pol = np.where(ib >= 0, (w[np.maximum(ib, 0)] >> 11) & 1, self.base_p)
        cum_at_base = np.where(ib >= 0, cum[np.maximum(ib, 0)], 0)
        vbase = base + (cum - adv - cum_at_base)
        vw = np.nonzero(adv > 0)[0]
        bits = (w[vw, None] >> np.arange(12)[None, :]) & 1
        bits[typ[vw] == T_VECT_8, 8:] = 0
        vr, vb = np.nonzero(bits)
        vi = vw[vr]
        # ---- merge in word order (a vector word's events in ascending column order) ----
        n_ev = len(sx) + len(vi)
        out = np.zeros(n_ev, EVENT_CD_DTYPE)
        order_key = np.concatenate((sx * 16, vi * 16 + vb))
        order = np.argsort(order_key, kind="stable")
        xs = np.concatenate((w[sx] & 0x7ff, vbase[vi] + vb))[order]
        ys = np.concatenate((y[sx], y[vi]))[order]
        ps = np.concatenate(((w[sx] >> 11) & 1, pol[vi]))[order]
        ts = np.concatenate((t[sx], t[vi]))[order]
        out["x"], out["y"], out["p"], out["t"] = xs, ys, ps, ts
        if self.wait_for_time_base and not self.have_time:
            src = np.concatenate((sx, vi))[order]            # the word each event came from
            out = out[ih[src] >= 0]                           # ... has a TIME_HIGH word at or before it in this chunk
        self.have_time = self.have_time or len(hi_words) > 0
        # ---- carry the state over to the next chunk ----
        self.y = int(y[-1])
        self.t_high, self.t_low = int(t_high[-1]), int(t_low[-1])
        self.t_loops = int(loops_at[-1])
        if ib[-1] >= 0 or len(vw):
            self.base_x = int(base[-1] + (cum[-1] - cum_at_base[-1]))
            self.base_p = int(pol[-1])
        return out


def decode_evt3(words: np.ndarray, wait_for_time_base: bool = False) -> np.ndarray:
    return Evt3Decoder(wait_for_time_base).decode(words)


class DeviceEvt3Decoder:
    """The same decoder as three kernels (csrc/xmaps_evt3.hpp: the state machine as scans): the words cross PCIe as the
    recording stores them, the records stay on the device.

        dec = DeviceEvt3Decoder(engine)
        ptr, n = dec.decode_device(words)       # 16-byte EventCD records in device memory, valid until the next call
        evs = dec.decode(words)                 # ... copied back (tests)
        n = dec.push(ingest, words)             # one chunk = one packet of a DeviceIngest (xm_ingest_push_evt3)

    State (row, time, vector base, 24-bit wraps) carries over from chunk to chunk, as in Evt3Decoder."""

    def __init__(self, engine, max_words: int = 1 << 20, max_events: int = 0, wait_for_time_base: bool = False):
        import ctypes as C

        from . import _native as N
        self._C, self._N, self._e = C, N, engine
        self._lib = engine._lib
        self._d = C.c_void_p(None)
        self.max_words = int(max_words)
        N.check(self._lib.xm_evt3_create(engine._h, int(max_words), int(max_events), C.byref(self._d)))
        if wait_for_time_base:  # (Evt3Decoder's option: events in front of the stream's first EVT_TIME_HIGH are not emitted)
            N.check(self._lib.xm_evt3_wait_for_time_base(self._d, 1))

    def close(self):
        if getattr(self, "_d", None) is not None and self._d.value:
            self._lib.xm_evt3_destroy(self._d)
            self._d = self._C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def reset(self):
        self._N.check(self._lib.xm_evt3_reset(self._d))

    def decode_device(self, words: np.ndarray):
        C = self._C
        w = np.ascontiguousarray(words, dtype="<u2")
        ptr, n = C.c_void_p(None), C.c_size_t(0)
        self._N.check(self._lib.xm_evt3_decode(self._d, C.c_void_p(w.ctypes.data), len(w), C.byref(ptr), C.byref(n)))
        return int(ptr.value or 0), int(n.value)

    def decode(self, words: np.ndarray) -> np.ndarray:
        out = []
        w = np.ascontiguousarray(words, dtype="<u2")
        for a in range(0, max(len(w), 1), self.max_words):
            ptr, n = self.decode_device(w[a:a + self.max_words])
            ev = np.zeros(n, EVENT_CD_DTYPE)
            if n:
                self._e.dev_download(ev, ptr)
            out.append(ev)
        cat = np.zeros(sum(len(e) for e in out), EVENT_CD_DTYPE)  # (np.concatenate hands back the packed 14-byte layout under NumPy 2)
        o = 0
        for e in out:
            cat[o:o + len(e)] = e
            o += len(e)
        return cat

    def push(self, ingest, words: np.ndarray, pinned: bool = False, count: bool = True):
        """One chunk = one packet of `ingest`.  pinned=True: `words` lies in pinned host memory (XMapsEngine.host_empty): no
        staging copy.  count=True: waits for the decoder and returns the chunk's event count; count=False: nothing is waited
        for (the ingest's kernels read the count on the device), returns None."""
        C = self._C
        w = np.ascontiguousarray(words, dtype="<u2")
        ingest._backpressure(1)
        if not count:
            self._N.check(self._lib.xm_ingest_push_evt3(ingest._g, self._d, C.c_void_p(w.ctypes.data), len(w), int(bool(pinned)), None))
            return None
        n = C.c_size_t(0)
        self._N.check(self._lib.xm_ingest_push_evt3(ingest._g, self._d, C.c_void_p(w.ctypes.data), len(w), int(bool(pinned)), C.byref(n)))
        return int(n.value)


def read_raw(path: str, chunk_words: int = 1 << 22):
    """Yields EventCD packets of a .raw file (EVT 3.0)."""
    with open(path, "rb") as f:
        blob = f.read()
    fields, off = split_raw_header(blob)
    fmt = fields.get("evt", fields.get("format", "3.0"))
    # "% evt 3.0" (older headers) or "% format EVT3;height=720;width=1280" (newer ones): the first token decides
    if fmt.split(";")[0].strip().upper() not in ("3.0", "3", "EVT3", "EVT3.0"):
        raise ValueError(f"{path}: only EVT 3.0 is supported (header says {fmt!r})")
    words = np.frombuffer(blob, dtype="<u2", offset=off, count=(len(blob) - off) // 2)
    dec = Evt3Decoder()
    for a in range(0, len(words), chunk_words):
        ev = dec.decode(words[a:a + chunk_words])
        if len(ev):
            yield ev


def read_raw_words(path: str, chunk_words: int = 1 << 20):
    """Yields the EVT 3.0 words of a .raw file chunk by chunk, undecoded: for DeviceEvt3Decoder / process_evt3_words."""
    with open(path, "rb") as f:
        blob = f.read()
    fields, off = split_raw_header(blob)
    fmt = fields.get("evt", fields.get("format", "3.0"))
    if fmt.split(";")[0].strip().upper() not in ("3.0", "3", "EVT3", "EVT3.0"):
        raise ValueError(f"{path}: only EVT 3.0 is supported (header says {fmt!r})")
    words = np.frombuffer(blob, dtype="<u2", offset=off, count=(len(blob) - off) // 2)
    for a in range(0, len(words), chunk_words):
        yield words[a:a + chunk_words]


def encode_evt3(evs: np.ndarray, use_vectors: bool = True) -> np.ndarray:
    """EventCD (time-ordered) -> EVT 3.0 words.  Test helper / file writer: runs of events that share (t, y, p) and have
    increasing columns within a 12-column window become VECT_BASE_X + VECT_12, everything else EVT_ADDR_X."""
    words = []
    cur_y = cur_hi = cur_lo = None
    x, y, p, t = (evs[k].astype(np.int64) for k in ("x", "y", "p", "t"))
    n, i = len(evs), 0
    while i < n:
        hi, lo = (t[i] >> 12) & 0xfff, t[i] & 0xfff
        if hi != cur_hi:
            words.append((T_TIME_HIGH << 12) | hi)
            cur_hi = hi
            cur_lo = None  # a TIME_HIGH that changes the high field restarts the low field at 0 in the decoders: always re-send it
        if lo != cur_lo:
            words.append((T_TIME_LOW << 12) | lo)
            cur_lo = lo
        if y[i] != cur_y:
            words.append((T_ADDR_Y << 12) | (y[i] & 0x7ff))
            cur_y = y[i]
        j = i + 1
        while (use_vectors and j < n and t[j] == t[i] and y[j] == y[i] and p[j] == p[i] and x[j] > x[j - 1]
               and x[j] - x[i] < 12):
            j += 1
        if j - i >= 2:
            mask = 0
            for k in range(i, j):
                mask |= 1 << int(x[k] - x[i])
            words.append((T_VECT_BASE_X << 12) | ((p[i] & 1) << 11) | (x[i] & 0x7ff))
            words.append((T_VECT_12 << 12) | mask)
        else:
            words.append((T_ADDR_X << 12) | ((p[i] & 1) << 11) | (x[i] & 0x7ff))
            j = i + 1
        i = j
    return np.array(words, dtype="<u2")


def encode_evt3_singles(evs: np.ndarray) -> np.ndarray:
    """EventCD (time-ordered) -> EVT 3.0 words without vector words, vectorised (the file writer above is a Python loop): per
    event [TIME_HIGH if the high field changed] [TIME_LOW if the low or the high field changed] [ADDR_Y if the row changed] ADDR_X."""
    n = len(evs)
    if n == 0:
        return np.zeros(0, "<u2")
    x, y, p, t = (evs[k].astype(np.int64) for k in ("x", "y", "p", "t"))
    hi, lo = (t >> 12) & 0xfff, t & 0xfff
    first = np.zeros(n, bool)
    first[0] = True
    c_hi = first | (np.concatenate(([0], hi[:-1])) != hi)
    c_lo = first | c_hi | (np.concatenate(([0], lo[:-1])) != lo)  # (a changed high field restarts the low field: re-send it)
    c_y = first | (np.concatenate(([0], y[:-1])) != y)
    per = c_hi.astype(np.int64) + c_lo + c_y + 1
    end = np.cumsum(per)  # one past the event's ADDR_X word
    words = np.zeros(int(end[-1]), np.int64)
    words[end - 1] = (T_ADDR_X << 12) | ((p & 1) << 11) | (x & 0x7ff)
    pos = end - 1 - c_y
    words[pos[c_y]] = ((T_ADDR_Y << 12) | (y & 0x7ff))[c_y]
    pos = pos - c_lo
    words[pos[c_lo]] = ((T_TIME_LOW << 12) | lo)[c_lo]
    pos = pos - c_hi
    words[pos[c_hi]] = ((T_TIME_HIGH << 12) | hi)[c_hi]
    return words.astype("<u2")


def write_raw(path: str, evs: np.ndarray, width: int = 640, height: int = 480):
    hdr = f"% evt 3.0\n% format EVT3;height={height};width={width}\n% geometry {width}x{height}\n% end\n".encode("ascii")
    with open(path, "wb") as f:
        f.write(hdr)
        f.write(encode_evt3(evs).tobytes())
