#!/usr/bin/env python3
"""LDS bank-conflict model of ed_cnn_mfma_kernel's operand reads and epilogue stores, and the search for the column orders the
kernel uses (csrc/cnn_mfma_cols.h).

Model (MI355X guide, LDS section): a ds_read_b128 is served in four 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, +32, one
LDS cycle per group plus one per extra distinct address on a busy bank, bank of byte address a = (a / 4) mod 64; a ds_write_b128 in
eight groups of 8 consecutive lanes, bank (a / 4) mod 32. Checked against the counters: the model's extra cycles per group of four
utterances for the natural column order (conv1 75, conv2 280, conv3 102, conv4 36) are what SQ_LDS_BANK_CONFLICT loses when the layer is
skipped (79 / 280 / 103 / 44, profiles/r04_cnn_lds_by_phase.txt).

Which column of a layer's GEMM a lane computes is free (every lane derives its own addresses), so the columns are dealt to the
(tile, lane) slots such that the 16 lanes of a read group hit 16 different 16-byte bank groups and the 8 lanes of a store group 8
different ones. `--search` runs the local search and prints the header; without it the header's tables are evaluated.
"""
import argparse, random, re, os, sys

UTT, REGA, INODD, P2PLANE, C3PLANE = 2992, 1120, 256, 560, 240
RG = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
RG = RG + [[l + 32 for l in g] for g in RG]
WG = [list(range(8 * i, 8 * i + 8)) for i in range(8)]


def cycles(addrs, groups, nbanks):
    tot = 0
    for g in groups:
        per_bank = {}
        for l in g:
            ad = addrs[l]
            if ad is None: continue
            for b in range(4):
                per_bank.setdefault(((ad // 4) + b) % nbanks, set()).add(ad)
        tot += max([len(v) for v in per_bank.values()] + [1])
    return tot


# ---- per layer: n live columns, tiles, and for a column q the read / store byte addresses of lane half h (within the wave's slice)
class Conv1:
    name, n, tiles, lanes = "conv1", 52, 2, 32
    @staticmethod
    def reads(q, h):
        u, py = divmod(q, 13)
        out = []
        for s in range(3):
            for odd in (0, 1):
                c = min(2 * s + h, 4) + odd
                out.append(u * UTT + py * 16 + (c & 1) * INODD + (c >> 1) * 16)
        return out
    @staticmethod
    def writes(q, h):
        u, py = divmod(q, 13)
        return [u * UTT + REGA + (py * 9 + 2 * rt + h) * 16 if 2 * rt + h < 9 else None for rt in range(5)]


class Conv2:
    name, n, tiles, lanes = "conv2", 140, 5, 32
    @staticmethod
    def reads(q, h):
        u, r = divmod(q, 35); py, x = divmod(r, 7)
        out = []
        for s in range(5):
            tap = min(2 * s + h, 8); ky, kx = divmod(tap, 3)
            for row in (0, 1):
                out.append(u * UTT + REGA + ((2 * py + ky + row) * 9 + x + kx) * 16)
        return out
    @staticmethod
    def writes(q, h):
        u, r = divmod(q, 35); py, x = divmod(r, 7)
        return [u * UTT + (py * 7 + x) * 16 + P2PLANE * h]


class Conv3:
    name, n, tiles, lanes = "conv3", 60, 2, 32
    @staticmethod
    def reads(q, h):
        u, r = divmod(q, 15); y, x = divmod(r, 5)
        return [u * UTT + P2PLANE * h + ((y + s // 3) * 7 + x + s % 3) * 16 for s in range(9)]
    @staticmethod
    def writes(q, h):
        u, r = divmod(q, 15); y, x = divmod(r, 5)
        return [u * UTT + REGA + (y * 5 + x) * 16 + C3PLANE * (half + h) for half in (0, 2)]


def layer_cost(L, cols):
    """cols[t][c] = live column or None; LDS cycles over the conflict-free count, reads and writes"""
    return sum(tile_cost(L, cols, t) for t in range(L.tiles))


def tile_cost(L, cols, t):
    extra = 0
    if True:
        last = max(q for q in cols[t] if q is not None)
        rd = [[L.reads(cols[t][l & 31] if cols[t][l & 31] is not None else last, l >> 5) for l in range(64)]]
        n_r = len(rd[0][0])
        for i in range(n_r):
            extra += cycles([rd[0][l][i] for l in range(64)], RG, 64) - 4
        wr = [L.writes(cols[t][l & 31], l >> 5) if cols[t][l & 31] is not None else None for l in range(64)]
        n_w = len(next(w for w in wr if w is not None))
        for i in range(n_w):
            extra += cycles([w[i] if w is not None else None for w in wr], WG, 32) - 8
    return extra


def natural(L):
    return [[(t * 32 + c if t * 32 + c < L.n else None) for c in range(32)] for t in range(L.tiles)]


def search(L, seed=1, iters=200000):
    rng = random.Random(seed)
    cols = natural(L)
    tc = [tile_cost(L, cols, t) for t in range(L.tiles)]
    cost = sum(tc)
    best, best_cost = [r[:] for r in cols], cost
    T = 2.0
    for it in range(iters):
        if best_cost == 0: break
        t1, c1, t2, c2 = rng.randrange(L.tiles), rng.randrange(32), rng.randrange(L.tiles), rng.randrange(32)
        if cols[t1][c1] is None and cols[t2][c2] is None: continue
        # every tile keeps at least one live column
        cols[t1][c1], cols[t2][c2] = cols[t2][c2], cols[t1][c1]
        if any(all(q is None for q in row) for row in cols):
            cols[t1][c1], cols[t2][c2] = cols[t2][c2], cols[t1][c1]; continue
        n1 = tile_cost(L, cols, t1)
        n2 = tile_cost(L, cols, t2) if t2 != t1 else n1
        c = cost - tc[t1] - (tc[t2] if t2 != t1 else 0) + n1 + (n2 if t2 != t1 else 0)
        if c <= cost or rng.random() < pow(2.718, (cost - c) / T):
            cost = c; tc[t1] = n1; tc[t2] = n2
            if c < best_cost: best, best_cost = [r[:] for r in cols], c
        else:
            cols[t1][c1], cols[t2][c2] = cols[t2][c2], cols[t1][c1]
        T = max(0.05, T * 0.99997)
    return best, best_cost


HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "edison_amd", "csrc", "cnn_mfma_cols.h")


def read_header():
    text = open(HEADER).read()
    out = {}
    for name in ("conv1", "conv2", "conv3"):
        m = re.search(r"ED_CNN_COLS_%s\[\d+\]\[32\] = \{(.*?)\};" % name.upper(), text, re.S)
        rows = re.findall(r"\{([^{}]*)\}", m.group(1))
        out[name] = [[(int(v) if int(v) >= 0 else None) for v in r.split(",")] for r in rows]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--search", action="store_true")
    ap.add_argument("--iters", type=int, default=150000)
    a = ap.parse_args()
    layers = (Conv1, Conv2, Conv3)
    if not a.search:
        tabs = read_header() if os.path.exists(HEADER) else {}
        for L in layers:
            line = "%-6s natural order: +%d LDS cycles per group over conflict-free" % (L.name, layer_cost(L, natural(L)))
            if L.name in tabs: line += "; csrc/cnn_mfma_cols.h: +%d" % layer_cost(L, tabs[L.name])
            print(line)
        return
    print("/* generated by tools/dev/cnn_lds_model.py --search: do not edit. Column of the layer's GEMM (conv1: utt * 13 + py; conv2: utt * 35 +\n"
          " * py * 7 + x; conv3: utt * 15 + y * 5 + x) that lane (tile, column lane) of a FULL group of four utterances computes, -1 = idle.\n"
          " * Chosen so that the ds_read_b128 groups of the operand reads and the ds_write_b128 groups of the epilogue stores meet no bank\n"
          " * conflict (LDS cycles per group over the conflict-free count, natural order -> this order: see the lines below). */")
    print("#ifndef ED_CNN_MFMA_COLS_H\n#define ED_CNN_MFMA_COLS_H")
    for L in layers:
        best, cost = search(L, iters=a.iters)
        assert sorted(q for r in best for q in r if q is not None) == list(range(L.n))
        print("/* %s: +%d -> +%d */" % (L.name, layer_cost(L, natural(L)), cost))
        print("static const short ED_CNN_COLS_%s[%d][32] = {" % (L.name.upper(), L.tiles))
        for r in best: print("\t{" + ", ".join(str(q if q is not None else -1) for q in r) + "},")
        print("};")
    print("#endif")


if __name__ == "__main__":
    main()
