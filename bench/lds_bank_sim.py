"""LDS bank-conflict model of gfx950 (lane groups and bank widths of MI355X_MICROARCH.md, "LDS" table)
and the exhaustive search that produced `s64w` in csrc/conv_first.hip: which GF(2)-linear XOR swizzle
of a 64-byte-row image ([rows][32 bf16], four 16-byte chunks) is conflict-free for
  (i)   ds_read_b128 fragment reads  (lane -> row lane & 15, logical chunk lane >> 4),
  (ii)  ds_write_b128 of the rows    (lane -> row, one logical chunk per instruction),
  (iii) ds_read_b64_tr_b16 reads     (lane (g4, q, pp) -> row 8*g4 + q (+4), chunk 2j + (pp >> 1), half pp & 1).
  python bench/lds_bank_sim.py"""
import itertools

G128 = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
        list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
        list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
        list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]


def ways(accesses, mod):
    """Worst number of DISTINCT addresses on one 4-byte bank among the accesses of one LDS cycle."""
    per = {}
    for addr, nbytes in accesses:
        for k in range(0, nbytes, 4):
            per.setdefault(((addr + k) // 4) % mod, set()).add(addr)
    return max(len(v) for v in per.values())


def s_of(row, M):
    b = [(row >> i) & 1 for i in range(4)]
    return (sum(M[0][i] & b[i] for i in range(4)) & 1) | ((sum(M[1][i] & b[i] for i in range(4)) & 1) << 1)


def score(M):
    r = max(ways([((l & 15) * 64 + (((l >> 4) ^ s_of(l & 15, M)) << 4), 16) for l in g], 64) for g in G128)
    w = max(ways([(row * 64 + ((ch ^ s_of(row & 15, M)) << 4), 16) for row in range(g0, g0 + 8)], 32)
            for ch in range(4) for g0 in range(0, 64, 8))
    t = 1
    for j in range(2):
        for hi in (0, 4):
            for half in (range(0, 32), range(32, 64)):
                acc = []
                for l in half:
                    g4, li = l >> 4, l & 15
                    q, pp = li >> 2, li & 3
                    row = 8 * g4 + q + hi
                    acc.append((row * 64 + (((2 * j + (pp >> 1)) ^ s_of(row & 15, M)) << 4) + (pp & 1) * 8, 8))
                t = max(t, ways(acc, 64))
    return r, w, t


if __name__ == "__main__":
    res = sorted((score((b[:4], b[4:])), (b[:4], b[4:])) for b in itertools.product([0, 1], repeat=8))
    print("(fragment reads, row writes, transposed reads) worst-case ways, then the map (chunk bit 0 row, chunk bit 1 row):")
    for sc, M in res[:6]:
        print(sc, M)
    print("forward kernel's (row >> 1) & 3:", score(((0, 1, 0, 0), (0, 0, 1, 0))))
    print("s64w = bit2 | (bit1 ^ bit2 ^ bit3) << 1:", score(((0, 0, 1, 0), (0, 1, 1, 1))))
