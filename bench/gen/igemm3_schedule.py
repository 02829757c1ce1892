"""Schedule generator + checker for the tap-sharing 8-phase implicit GEMM (igemm.hip, PIPE == 3).

K-tile T = cc*9 + tap (chunk-major), tap = 3*(dy+1) + (dx+1).  The three dx taps of a (cc, dy) GROUP
read ONE shared activation image (258 pixel rows, fragment reads at row offsets -1/0/+1), so the
activation half of the LDS fill drops to a third.  Per lane and phase this script lists what is
issued (B half-tiles: NB loads; A-image pieces: 1 load) and derives, for the 24-phase loop body
(6 K-tiles = lcm(2 B buffers, 3 taps per group)), the counted `s_waitcnt vmcnt(N)` of every phase:
N = loads issued after the last load that the NEXT phase's reads need.  A simulation over many tiles
checks RAW (needed loads complete under the in-order vmcnt rule, given a full drain after the
prologue) and WAR (a slot is re-staged >= 2 phases after its last read).
Run: python bench/gen/igemm3_schedule.py  -> prints the N table used in igemm.hip."""
NB = 2          # loads per lane for a B half-tile (256x256: WGN / 2)
PIECES = 5      # A-image pieces per wave and group (33 needed over 8 waves, padded to 40)


def phase_plan(phi):
    """What phase phi (global index, tile = phi // 4) issues and what the NEXT phase reads."""
    t, q = divmod(phi, 4)
    issues = []      # (name, loads)
    if q == 0:
        issues.append((("B1", t + 1), NB))
    if q == 2:
        issues.append((("B0", t + 2), NB))
    g, tg = divmod(t, 3)            # group, tile within group
    k = tg * 4 + q                  # phase within the group
    if k < PIECES:
        issues.append((("A", g + 1, k), 1))
    return issues


def reads_of(phi):
    t, q = divmod(phi, 4)
    g = t // 3
    if q == 0:
        return [("B0", t)] + [("A", g, k) for k in range(PIECES)]
    if q == 1:
        return [("B1", t)]
    if q == 2:
        return [("A", g, k) for k in range(PIECES)]
    return []


def slot_of(name):
    if name[0] == "A":
        return ("A", name[1] & 1)
    return (name[0], name[1] & 1)


def derive(nphi=24 * 6):
    stream = []                      # (name, phase) one entry per LOAD, in issue order
    table = {}
    issued_at = {}
    for phi in range(nphi):
        for name, n in phase_plan(phi):
            for _ in range(n):
                stream.append((name, phi))
            issued_at[name] = phi
        need = [x for x in reads_of(phi + 1) if x in issued_at]
        if not need or phi < 24:     # warm-up region: steady state from phase 24 on
            pass
        last = max((i for i, (nm, _) in enumerate(stream) if nm in need), default=None)
        n_young = len(stream) - 1 - last if last is not None else None
        if phi >= 24 and phi < nphi - 24:
            key = phi % 24
            if n_young is not None:
                table.setdefault(key, set()).add(n_young)
    out = {}
    for key in range(24):
        vals = table.get(key)
        out[key] = min(vals) if vals else None     # None: nothing new needed -> reuse the previous phase's N
    return out


def simulate(table, ntiles=36):
    """In-order completion model: at a wait vmcnt(N) everything except the N youngest loads is complete."""
    stream, complete_upto = [], 0
    last_read_phase = {}
    # prologue: A image of group 0, B0(0), B1(0), B0(1), then a full drain
    for k in range(PIECES):
        stream.append(("A", 0, k))
    for nm in (("B0", 0), ("B1", 0), ("B0", 1)):
        stream.extend([nm] * NB)
    complete_upto = len(stream)
    done = set(stream)
    for phi in range(ntiles * 4):
        for x in reads_of(phi):                      # reads at the start of the phase
            assert x in done, "RAW: phase %d reads %r before it is complete" % (phi, x)
            last_read_phase[slot_of(x)] = phi
        for name, n in phase_plan(phi):
            s = slot_of(name)
            if s in last_read_phase:
                assert phi - last_read_phase[s] >= 2, "WAR: phase %d restages %r read in phase %d" % (phi, s, last_read_phase[s])
            stream.extend([name] * n)
        n = table[phi % 24]
        complete_upto = max(complete_upto, len(stream) - n)
        done = set(stream[:complete_upto])
    return True


if __name__ == "__main__":
    t = derive()
    prev = None
    filled = {}
    for k in range(24):
        v = t[k]
        if v is None:
            v = prev + sum(n for _, n in phase_plan(24 + k))   # nothing new needed: keep the same loads waited for
        filled[k] = v
        prev = v
    # second pass for phase 0 depending on phase 23
    print("vmcnt per phase of the 24-phase body:", [filled[k] for k in range(24)])
    assert simulate(filled)
    print("simulation OK (RAW + WAR) over 36 K-tiles")
    for k in range(24):
        print(k, phase_plan(24 * 3 + k), "next reads", reads_of(24 * 3 + k + 1)[:2])
