"""numpy model of the backward scan's "xor scatter" reduction (csrc/scan_bwd.hip::xor_reduce16 + rev_walk): lane l keeps state
k ^ h(l) in register slot k; four DPP levels and two swap levels leave, in every lane, one finished 64-lane total.
Run on the CPU (also imported by tests/test_host_logic.py): checks every lane's (array, state, step) against plain sums."""
import numpy as np

L = np.arange(64)


def hmask(l):
    b0, b1, b2 = l & 1, (l >> 1) & 1, (l >> 2) & 1
    return (b0 ^ b2) | ((b1 ^ b2) << 1) | (b2 << 2)


H = hmask(L)


def qp1(v): return v[L ^ 1]                               # quad_perm:[1,0,3,2]
def qp2(v): return v[L ^ 2]                               # quad_perm:[2,3,0,1]
def half_mirror(v): return v[(L & ~7) | (7 - (L & 7))]    # row_half_mirror
def ror8(v): return v[(L & ~15) | ((L + 8) & 15)]         # row_ror:8


def step_totals(pb, pc):
    """pb, pc: [state][lane] contributions of one time step -> X[lane] (xor_reduce16)."""
    b = [pb[k ^ H, L] for k in range(8)]                  # slot k of lane l holds state k ^ h(l)
    c = [pc[k ^ H, L] for k in range(8)]
    for a in (b, c):
        for k in (0, 2, 4, 6):
            a[k] = a[k] + qp1(a[k + 1])
        for k in (0, 4):
            a[k] = a[k] + qp2(a[k + 2])
        a[0] = a[0] + half_mirror(a[4])
    return np.where(((L >> 3) & 1) == 0, b[0] + ror8(b[0]), c[0] + ror8(c[0]))


def swap32_add(a, b):                                      # v_permlane32_swap: lanes 32-63 of a <-> lanes 0-31 of b
    a2, b2 = a.copy(), b.copy()
    a2[32:], b2[:32] = b[:32], a[32:]
    return a2 + b2


def swap16_add(a, b):                                      # v_permlane16_swap: odd rows of a <-> even rows of b
    a2, b2 = a.copy(), b.copy()
    for r in (1, 3):
        a2[16 * r:16 * r + 16] = b[16 * (r - 1):16 * r]
        b2[16 * (r - 1):16 * r] = a[16 * r:16 * r + 16]
    return a2 + b2


def half_walk(PB, PC):
    """PB, PC: [step 0..7][state][lane] -> {base step: Z[lane]} as rev_walk forms them (steps walked 7 .. 0)."""
    out, Xo, Yo = {}, None, None
    for s in range(7, -1, -1):
        X = step_totals(PB[s], PC[s])
        if s & 1:
            Xo = X
        else:
            Y = swap32_add(Xo, X)
            if s & 2:
                Yo = Y
            else:
                out[s] = swap16_add(Yo, Y)
    return out


def lane_owner(l, base):
    """(array, state, step) whose total lane l of the register stored at `base` holds."""
    r = l >> 4
    return (l >> 3) & 1, int(hmask(l & 7)), base + 3 - (2 * (r & 1) + (r >> 1))


def check(seed=0):
    rng = np.random.default_rng(seed)
    PB, PC = rng.standard_normal((8, 8, 64)), rng.standard_normal((8, 8, 64))
    Z = half_walk(PB, PC)
    seen = set()
    for base, z in Z.items():
        for l in range(64):
            arr, n, st = lane_owner(l, base)
            ref = (PC if arr else PB)[st, n, :].sum()
            assert abs(z[l] - ref) < 1e-9, (l, base, z[l], ref)
            seen.add((arr, n, st))
    assert len(seen) == 2 * 8 * 8                       # every (array, state, step) total has exactly one owner lane
    # the staged B / C variants: position q of variant v holds state q ^ v, rotated by 16 v floats; a lane reads the float4
    # at n0 + 4 * (j ^ h2) of variant h & 3 and must see states n0 + (k ^ h), k = 4 j + i
    row = rng.standard_normal(64)
    tiles = np.zeros((4, 64))
    for v in range(4):
        for n in range(64):
            tiles[v, ((n ^ v) + 16 * v) & 63] = row[n]
    for n0 in range(0, 64, 8):
        for l in range(64):
            h = int(H[l])
            base = (n0 + 16 * (h & 3)) & 63
            for j in range(2):
                f4 = tiles[h & 3, base + 4 * (j ^ (h >> 2)):base + 4 * (j ^ (h >> 2)) + 4]
                for i in range(4):
                    assert f4[i] == row[n0 + ((4 * j + i) ^ h)]
    # bank groups: the eight distinct float4 of a wave's step sit on eight different 4-bank groups
    for n0 in range(0, 64, 8):
        groups = {(((n0 + 16 * v) & 63) + 4 * j) // 4 % 16 for v in range(4) for j in range(2)}
        assert len(groups) == 8
    return True


if __name__ == "__main__":
    print("xor scatter model:", "ok" if check() else "FAILED")
