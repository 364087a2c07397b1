#!/usr/bin/env python3
"""Enumerates the LDS bank conflicts of conv_b3_s2fir.hip's operand reads (ds_read_b128) and image stores (ds_write_b64) against the
lane groups of MI355X_MICROARCH.md (LDS): a ds_read_b128 is served in four groups of 16 lanes, bank = (addr / 4) mod 64."""
PITCH, ODD0, ROWB = 40, 24, 32
GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
GROUPS += [[l + 32 for l in g] for g in GROUPS]


def slot_byte(slot, half):
    return slot * ROWB + ((half ^ ((slot >> 3) & 1)) << 4)


def read_conflicts(pitch=PITCH, odd0=ODD0):
    worst = 0
    for rbg in range(4):
        for tap in range(9):
            ty, tx = divmod(tap, 3)
            addr = {}
            for lane in range(64):
                li, lh = lane & 31, lane >> 5
                pr, c = 2 * rbg + (li >> 4), li & 15
                slot = (2 * pr + ty) * pitch + (odd0 + c if tx == 1 else c + (tx >> 1))
                addr[lane] = slot_byte(slot, lh)
            for g in GROUPS:
                banks = {}
                for l in g:
                    for d in range(4):
                        banks.setdefault((addr[l] // 4 + d) % 64, set()).add(addr[l] + 4 * d)
                worst = max(worst, max(len(v) for v in banks.values()))
    return worst


def write_conflicts():
    """ds_write_b64 of the FIR stage: 4 x 16 contiguous lanes, bank = (addr / 4) mod 32 (the write path's banking)."""
    worst = 0
    for base in range(0, 448, 16):
        for col in (0, 1):
            for j in range(3):
                banks = {}
                for t in range(base, base + 16):
                    if t >= 408:
                        continue
                    quad, q, seg = t & 3, (t >> 2) % 17, (t >> 2) // 17
                    if col == 1 and q == 16:
                        continue
                    slot = (seg * 3 + j) * PITCH + (ODD0 + q if col else q)
                    a = slot * ROWB + ((quad * 8) ^ (((slot >> 3) & 1) << 4))
                    for d in range(2):
                        banks.setdefault((a // 4 + d) % 32, set()).add(a + 4 * d)
                if banks:
                    worst = max(worst, max(len(v) for v in banks.values()))
    return worst


if __name__ == "__main__":
    print("ds_read_b128 operand fetch, worst distinct addresses per bank and lane group:", read_conflicts(), "(1 = conflict-free)")
    for pitch in (33, 34, 36, 40, 48):
        print("  pitch", pitch, "->", read_conflicts(pitch))
    print("ds_write_b64 image store, worst per bank and 16-lane group:", write_conflicts())
