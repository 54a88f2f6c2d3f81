#!/usr/bin/env python3
"""LDS bank-conflict model of the BCJR pass (csrc/bcjr.hip) per the lane-group / bank rules of
/opt/skills/guides/MI355X_MICROARCH.md (section LDS): counts extra LDS-array cycles per wave-instruction for the accesses of
one phase-2 chunk, for the round-3 layout and the round-4 one.  Pure host arithmetic; used to choose row strides and the
item -> lane permutation before measuring SQ_LDS_BANK_CONFLICT on the GPU."""
import itertools
import sys

G_R64 = [list(range(0, 32)), list(range(32, 64))]
G_R128 = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
G_R128 = G_R128 + [[x + 32 for x in g] for g in G_R128]
G_W64 = [list(range(i, i + 16)) for i in range(0, 64, 16)]
G_W128 = [list(range(i, i + 8)) for i in range(0, 64, 8)]


def extra_cycles(addrs, groups, width, nbanks):
    """addrs[lane] = byte address (None = inactive); returns extra cycles (conflicts) of one wave-instruction"""
    extra = 0
    for grp in groups:
        banks = {}
        for ln in grp:
            a = addrs[ln]
            if a is None:
                continue
            for d in range(width // 4):
                banks.setdefault(((a // 4) + d) % nbanks, set()).add((a // 4 + d))
        worst = max((len(v) for v in banks.values()), default=1)
        extra += worst - 1
    return extra


def chunk(layout, GW=16, S=4, CH=8, verbose=False):
    """layout: dict with functions tab_g(tl, g, c) / tab_p(tl, g, i) -> double index; xs(tl, lane, slot); item(lane, q) -> (gg, tl);
    xrow(tl, gg, st) -> (double index of app0 term pair start, width)"""
    tot = {}

    def add(name, n_instr, extra):
        t = tot.setdefault(name, [0, 0])
        t[0] += n_instr
        t[1] += extra
    lanes = range(64)
    # stage stores: 2 items x (gamma 2 x b128 + prior b128)
    for q in range(2):
        items = [layout["item"](ln, q) for ln in lanes]
        for part in range(3):
            addrs = []
            for (gg, tl) in items:
                if gg >= GW:
                    addrs.append(None)
                    continue
                addrs.append(8 * (layout["tab_g"](tl, gg, 2 * part) if part < 2 else layout["tab_p"](tl, gg, 0)))
            add("stage store b128", 1, extra_cycles(addrs, G_W128, 16, 32))
    # recursion reads: per step 2 gamma + 2 prior b64 reads for alpha and for beta; code offsets per state: use a typical table
    # (4-state RSC (1, 5/7)): codes differ per state; worst case modelled with off = state-dependent 0..3
    for tl in range(CH):
        for which in range(2):                      # alpha_w, beta_w
            for br in range(2):
                addrs = [8 * layout["tab_g"](tl, ln >> 2, ((ln & 3) * 2 + br + which) % 4) for ln in lanes]
                add("gamma read b64", 1, extra_cycles(addrs, G_R64, 8, 64))
                addrs = [8 * layout["tab_p"](tl, ln >> 2, ((ln & 1) ^ br)) for ln in lanes]
                add("prior read b64", 1, extra_cycles(addrs, G_R64, 8, 64))
    # X stores: per step two b64 stores (slot = input of the branch, per lane)
    for tl in range(CH):
        for br in range(2):
            addrs = [8 * layout["xs"](tl, ln, ((ln >> 1) & 1) ^ br) for ln in lanes]
            add("x store b64", 1, extra_cycles(addrs, G_W64, 8, 32))
    # epilogue reads
    for q in range(2):
        items = [layout["item"](ln, q) for ln in lanes]
        for r in range(layout["xreads"]):
            addrs = [8 * layout["xread"](tl, gg, r) for (gg, tl) in items]
            add("epilogue read b128", 1, extra_cycles(addrs, G_R128, 16, 64))
    n = sum(v[0] for v in tot.values())
    e = sum(v[1] for v in tot.values())
    if verbose:
        for k, v in tot.items():
            print("  %-22s %3d instr, %4d extra cycles" % (k, v[0], v[1]))
        print("  total %d LDS instructions, %d conflict cycles = %.2f per instruction" % (n, e, e / n))
    return e, n


def round3():
    P, XS_ROW = 16 * 6 + 2, 132
    return {
        "tab_g": lambda tl, g, c: tl * P + g * 6 + c,
        "tab_p": lambda tl, g, i: tl * P + g * 6 + 4 + i,
        "xs": lambda tl, lane, slot: tl * XS_ROW + lane * 2 + slot,
        "item": lambda lane, q: ((lane + 64 * q) // 8, (lane + 64 * q) % 8),
        "xreads": 4, "xread": lambda tl, gg, r: tl * XS_ROW + gg * 8 + r * 2,
    }


def round4(GS=66, PS=34, XR=130, perm=(0, 4, 6, 2, 1, 5, 7, 3)):
    # xs slot-major: [tl][slot][64 lanes]; epilogue item (gg, tl) reads doubles gg*4 .. gg*4+3 of slot 0 and of slot 1
    return {
        "tab_g": lambda tl, g, c: tl * GS + g * 4 + c,
        "tab_p": lambda tl, g, i: 8 * GS + tl * PS + g * 2 + i,
        "xs": lambda tl, lane, slot: tl * XR + slot * 64 + lane,
        "item": lambda lane, q: (perm[lane >> 3] + 8 * q, lane & 7),
        "xreads": 4, "xread": lambda tl, gg, r: tl * XR + (r >> 1) * 64 + gg * 4 + (r & 1) * 2,
    }


if __name__ == "__main__":
    print("round 3 layout:")
    chunk(round3(), verbose=True)
    best = None
    for XR in range(128, 140, 2):
        for perm in itertools.permutations(range(8)):
            if perm[0] != 0:
                continue
            lay = round4(XR=XR, perm=perm)
            e = 0
            for q in range(2):
                items = [lay["item"](ln, q) for ln in range(64)]
                for r in range(4):
                    e += extra_cycles([8 * lay["xread"](tl, gg, r) for (gg, tl) in items], G_R128, 16, 64)
            if best is None or e < best[0]:
                best = (e, XR, perm)
        if best[0] == 0:
            break
    print("best epilogue mapping: extra %d, XS_ROW %d, perm %s" % best)
    print("round 4 layout:")
    chunk(round4(XR=best[1], perm=best[2]), verbose=True)
