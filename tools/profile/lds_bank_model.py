#!/usr/bin/env python3
"""LDS bank-conflict model of the decode kernel's accesses (dequant_idct_kernel, jpezy_kernels.hip), per the lane groups and
bank functions of /opt/skills/guides/MI355X_MICROARCH.md (section LDS): a wave64 access is served in fixed lane groups, one
LDS cycle per group when conflict-free; every extra distinct address on a busy bank adds a cycle.  Prints LDS-array cycles
per wave for each access class of the current geometry and of candidate geometries; the counters to compare with are
SQ_LDS_IDX_ACTIVE (all cycles) and SQ_LDS_BANK_CONFLICT (extra cycles).

    python tools/profile/lds_bank_model.py
"""
import itertools

G32 = [list(range(0, 32)), list(range(32, 64))]
G16C = [list(range(16 * g, 16 * g + 16)) for g in range(4)]
G8C = [list(range(8 * g, 8 * g + 8)) for g in range(8)]
G128R = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
G128R = G128R + [[l + 32 for l in g] for g in G128R]

KINDS = {  # name: (lane groups, bank modulus, dwords per lane)
    "ds_read_b16/b32": (G32, 32, 1), "ds_read_b64": (G32, 64, 2), "ds_read_b128": (G128R, 64, 4),
    "ds_write_b16/b32": (G32, 32, 1), "ds_write_b64": (G16C, 32, 2), "ds_write_b128": (G8C, 32, 4),
}


def cycles(kind, addr_bytes, active=None):
    """LDS-array cycles of one wave instruction; addr_bytes[lane] = byte address (None / inactive lanes skipped)"""
    groups, mod, width = KINDS[kind]
    total, ideal = 0, 0
    for g in groups:
        banks = {}
        any_active = False
        for lane in g:
            if active is not None and not active[lane]:
                continue
            a = addr_bytes[lane]
            if a is None:
                continue
            any_active = True
            for k in range(width):
                dw = a // 4 + k
                banks.setdefault(dw % mod, set()).add(dw)
        if any_active:
            total += max(len(s) for s in banks.values())
            ideal += 1
    return total, ideal


ZZ = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49,
      56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]
ZZINV = [0] * 64
for n, v in enumerate(ZZ):
    ZZINV[v] = n


def decode_model(stage="zigzag", blk_pitch=128, dh_pitch=20, dh_mcu=336, c_pitch=20, c_comp=160, c_mcu=336):
    """returns {access class: (cycles, conflict-free cycles)} per wave"""
    out = {}
    lanes = range(64)
    row = [l >> 2 for l in lanes]
    m = [l & 3 for l in lanes]
    # A. staging writes
    tot = ide = 0
    if stage == "zigzag":          # three lane-linear 16-byte stores (block pitch blk_pitch)
        for k in range(3):
            addr = [((k * 64 + l) >> 3) * blk_pitch + ((k * 64 + l) & 7) * 16 for l in lanes]
            c, i = cycles("ds_write_b128", addr)
            tot += c; ide += i
    else:                          # natural order: 24 two-byte stores, lane holds zig-zag positions 8p..8p+7 of block c>>3
        for k in range(3):
            for e in range(8):
                addr = [((k * 64 + l) >> 3) * blk_pitch + 2 * ZZ[8 * (l & 7) + e] for l in lanes]
                c, i = cycles("ds_write_b16/b32", addr)
                tot += c; ide += i
    out["stage write"] = (tot, ide)
    # B. column reads: lane (row, m) reads coefficient (v, u = row & 7) of blocks m*6 + {bx, 2 + bx, 4 + bx}, bx = row >> 3
    tot = ide = 0
    for blk_of in (lambda l: m[l] * 6 + (row[l] >> 3), lambda l: m[l] * 6 + 2 + (row[l] >> 3), lambda l: m[l] * 6 + 4 + (row[l] >> 3)):
        for v in range(8):
            if stage == "zigzag":
                addr = [blk_of(l) * blk_pitch + 2 * ZZINV[v * 8 + (row[l] & 7)] for l in lanes]
            else:
                addr = [blk_of(l) * blk_pitch + 2 * (v * 8 + (row[l] & 7)) for l in lanes]
            c, i = cycles("ds_read_b16/b32", addr)
            tot += c; ide += i
    out["column reads"] = (tot, ide)
    # C. luma half tiles: lanes with (row >> 3) == half write column u = row & 7 of rows y = 0..15 (doubles)
    tot = ide = 0
    for half in range(2):
        act = [(row[l] >> 3) == half for l in lanes]
        for y in range(16):
            addr = [4 * (m[l] * dh_mcu + (row[l] & 7) * 2 + y * dh_pitch) for l in lanes]
            c, i = cycles("ds_write_b64", addr, act)
            tot += c; ide += i
    out["luma tile writes"] = (tot, ide)
    # D. row reads: lane (row, m) reads 4 x double2 of its row
    tot = ide = 0
    for half in range(2):
        for k in range(4):
            addr = [4 * (m[l] * dh_mcu + row[l] * dh_pitch + 4 * k) for l in lanes]
            c, i = cycles("ds_read_b128", addr)
            tot += c; ide += i
    out["luma row reads"] = (tot, ide)
    # E. chroma tile
    tot = ide = 0
    for y in range(8):
        addr = [4 * (m[l] * c_mcu + (row[l] >> 3) * c_comp + (row[l] & 7) * 2 + y * c_pitch) for l in lanes]
        c, i = cycles("ds_write_b64", addr)
        tot += c; ide += i
    out["chroma tile writes"] = (tot, ide)
    tot = ide = 0
    for k in range(4):
        addr = [4 * (m[l] * c_mcu + (row[l] & 1) * c_comp + (row[l] >> 1) * c_pitch + 4 * k) for l in lanes]
        c, i = cycles("ds_read_b128", addr)
        tot += c; ide += i
    out["chroma row reads"] = (tot, ide)
    return out


def show(title, res):
    tot = sum(c for c, _ in res.values())
    ide = sum(i for _, i in res.values())
    print(f"{title}: {tot} LDS cycles per wave, {tot - ide} of them conflicts ({100 * (tot - ide) / tot:.0f} %)")
    for k, (c, i) in res.items():
        print(f"    {k:20s} {c:4d} cycles (conflict-free {i})")


if __name__ == "__main__":
    show("round-1 geometry (zig-zag staging, 128-byte blocks, MCU stride 336)", decode_model())
    show("144-byte block pitch", decode_model(blk_pitch=144))
    best = None
    for dh_mcu, c_mcu in itertools.product(range(320, 400, 4), range(320, 400, 4)):
        r = decode_model(stage="natural", blk_pitch=144, dh_mcu=dh_mcu, c_mcu=c_mcu)
        t = sum(c for c, _ in r.values())
        if best is None or t < best[0]:
            best = (t, dh_mcu, c_mcu)
    print("best MCU strides for the transposed tiles:", best)
    show(f"natural-order staging, 144-byte blocks, MCU strides {best[1]} / {best[2]}", decode_model(stage="natural", blk_pitch=144, dh_mcu=best[1], c_mcu=best[2]))
    show(f"zig-zag staging kept, MCU strides {best[1]} / {best[2]}", decode_model(dh_mcu=best[1], c_mcu=best[2]))
