"""Test helper: a small baseline-JPEG writer for layouts libjpeg (PIL) does not produce -- sampling factors up to 4,
factors that do not divide hmax/vmax, one-component files with H,V > 1, SOF0 precision != 8.  The files are not pictures:
the coefficients are random, the point is a valid entropy-coded stream in a chosen MCU layout that the oracle's restatement
of the reference decoder (read_jpeg / decode_mcu / make_rgb, ref decoder/jpezy_decoder.hpp:171-676) and the product both
decode.  Huffman tables: Annex K (the ones every encoder here writes), taken from a file PIL produces.
"""
import io
import struct

import numpy as np

ZZ = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21,
      28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54,
      47, 55, 62, 63]


def annex_k_tables():
    """{(tc, th): (bits[16], vals)} of the four standard tables, parsed from the DHT segments of a libjpeg file"""
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(np.zeros((8, 8, 3), np.uint8)).save(buf, "JPEG", quality=75)
    d = buf.getvalue()
    out, i = {}, 2
    while i < len(d):
        assert d[i] == 0xFF
        mk, n = d[i + 1], struct.unpack(">H", d[i + 2:i + 4])[0]
        if mk == 0xDA:
            break
        if mk == 0xC4:
            j, end = i + 4, i + 2 + n
            while j < end:
                tc, th = d[j] >> 4, d[j] & 15
                bits = list(d[j + 1:j + 17])
                nv = sum(bits)
                out[(tc, th)] = (bits, list(d[j + 17:j + 17 + nv]))
                j += 17 + nv
        i += 2 + n
    assert len(out) == 4
    return out


def canonical(bits, vals):
    code, k, tab = 0, 0, {}
    for ln in range(1, 17):
        for _ in range(bits[ln - 1]):
            tab[vals[k]] = (code, ln)
            code += 1
            k += 1
        code <<= 1
    return tab


class _Bits:
    def __init__(self):
        self.out = bytearray()
        self.acc = 0
        self.n = 0

    def put(self, code, ln):
        self.acc = (self.acc << ln) | (code & ((1 << ln) - 1))
        self.n += ln
        while self.n >= 8:
            b = (self.acc >> (self.n - 8)) & 0xFF
            self.out.append(b)
            if b == 0xFF:
                self.out.append(0)
            self.n -= 8
        self.acc &= (1 << self.n) - 1

    def flush(self):
        if self.n:
            self.put((1 << (8 - self.n)) - 1, 8 - self.n)          # pad with ones (T.81 F.1.2.3)


def _cat(v):
    return int(abs(int(v))).bit_length()


def synth_jpeg(W, H, comps, seed=0, density=0.15, amp=30, precision=8, qmax=24):
    """comps: [(H, V, Tq, Td)] for 1 or 3 components (Td selects BOTH the DC and the AC table in the reference, :630, so
    Ta = Td is written).  Returns (bytes, coeffs[mcu][block][64] zig-zag int16, info dict)."""
    rng = np.random.default_rng(seed)
    tabs = annex_k_tables()
    enc = {k: canonical(*v) for k, v in tabs.items()}
    hmax = max(c[0] for c in comps)
    vmax = max(c[1] for c in comps)
    hb, vb = (W + 7) // 8, (H + 7) // 8                             # get_blocks, ref :166-169
    mcu_cols, mcu_rows = -(-hb // hmax), -(-vb // vmax)
    bpm = sum(c[0] * c[1] for c in comps)
    nmcu = mcu_cols * mcu_rows
    co = np.zeros((nmcu, bpm, 64), np.int16)
    mask = rng.random(co.shape) < density
    co[mask] = rng.integers(-amp, amp + 1, int(mask.sum()), dtype=np.int16)
    co[..., 0] = rng.integers(-60, 61, (nmcu, bpm), dtype=np.int16)
    qt = rng.integers(1, qmax + 1, (2, 64), dtype=np.int64)

    o = bytearray(b"\xFF\xD8")
    for t in range(2):
        o += b"\xFF\xDB" + struct.pack(">HB", 67, t) + bytes(int(qt[t][ZZ[i]]) for i in range(64))
    o += b"\xFF\xC0" + struct.pack(">HBHHB", 8 + 3 * len(comps), precision, H, W, len(comps))
    for i, (h, v, tq, _) in enumerate(comps):
        o += bytes([i + 1, (h << 4) | v, tq])
    for (tc, th), (bits, vals) in sorted(tabs.items()):
        o += b"\xFF\xC4" + struct.pack(">HB", 19 + len(vals), (tc << 4) | th) + bytes(bits) + bytes(vals)
    o += b"\xFF\xDA" + struct.pack(">HB", 6 + 2 * len(comps), len(comps))
    for i, (_, _, _, td) in enumerate(comps):
        o += bytes([i + 1, (td << 4) | td])
    o += b"\x00\x3F\x00"

    bw = _Bits()
    pred = [0] * len(comps)
    for m in range(nmcu):
        b = 0
        for ci, (h, v, _, td) in enumerate(comps):
            dc_t, ac_t = enc[(0, td)], enc[(1, td)]
            for _ in range(h * v):
                z = co[m, b]
                b += 1
                diff = int(z[0]) - pred[ci]
                pred[ci] = int(z[0])
                s = _cat(diff)
                bw.put(*dc_t[s])
                if s:
                    bw.put(diff if diff >= 0 else diff - 1, s)
                run = 0
                for k in range(1, 64):
                    a = int(z[k])
                    if a == 0:
                        run += 1
                        continue
                    while run > 15:
                        bw.put(*ac_t[0xF0])
                        run -= 16
                    s = _cat(a)
                    bw.put(*ac_t[(run << 4) | s])
                    bw.put(a if a >= 0 else a - 1, s)
                    run = 0
                if run:
                    bw.put(*ac_t[0x00])
    bw.flush()
    o += bw.out + b"\xFF\xD9"
    info = dict(W=W, H=H, hmax=hmax, vmax=vmax, mcu_cols=mcu_cols, mcu_rows=mcu_rows, bpm=bpm, qt=qt)
    return bytes(o), co.reshape(-1), info
