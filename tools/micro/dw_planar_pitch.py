"""Exhaustive search behind DwTile::PL / row_pitch of csrc/dwconv.hip (DW_LDS_MODE 2): smallest (row pitch, plane pitch) for which every
16-lane ds_read_b128 group of the stencil's read pattern covers the 64 LDS banks once.  Lane groups: MI355X_MICROARCH.md (LDS table)."""
G = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
G = G + [[l + 32 for l in g] for g in G]


def worst(PL, RP, STRIPS, R=10):
    w = 0
    for g in G:
        banks = {}
        for l in g:
            cg, strip, y = l % 4, (l // 4) % STRIPS, l // (4 * STRIPS)
            a = y * RP + cg * PL + strip * R * 8
            for b in range(4):
                banks.setdefault(((a // 4) + b) % 64, set()).add(a)
        w = max(w, max(len(v) for v in banks.values()))
    return w


if __name__ == "__main__":
    for TW, STRIPS in ((40, 4), (20, 2)):
        min_pl = (TW + 14) * 8          # planes hold TW + k - 1 pixels of 8 bytes, k <= 15
        sols = sorted((RP, PL) for PL in range((min_pl + 15) // 16 * 16, 640, 16) for RP in range(4 * PL, 4 * PL + 512, 16) if worst(PL, RP, STRIPS) == 1)
        print(f"{TW}-wide tile: conflict-free (row pitch, plane pitch) = {sols[:4]}")
