"""A numpy walk of the matrix-core PLAN of a graph (edison_amd/csrc/model_net_mm.c), byte for byte as ed_net_mfma_kernel
walks it on the GPU: the input table, the activation region with its in / out offsets, expansion records, chunk offsets,
column tables, operand fragments, seeds, fused pooling windows, requantisation, the consumer layouts (zero borders, pixel
gaps, Toeplitz rows). Test infrastructure: it lets the CPU suite check the HOST side of the general network path -- the
planner -- against oracle/net_ref.py without a GPU. Reads of LDS bytes nothing wrote meet a poison value, indices outside
the wave's slice raise.

The struct layouts come from the library itself (edison_net_plan_layout); the field ORDER is restated here from
csrc/edison_internal.h (all fields are int32)."""
import ctypes

import numpy as np

NET_LAYER = "type relu in_h in_w in_c out_h out_w out_c kh kw sh sw pad_h pad_w rs w_off seed_off in_buf out_buf in_n out_n acts_off check_taps pad_".split()
NET_HEAD = "n_layers in_h in_w in_c in_n out_n logits_layer has_softmax lds_bytes acts_bytes weights_bytes n_seeds".split()
MM_LAYER = ("mm in_hp in_wp in_py in_px in_img expand x_img cpr n_ks n_rt pitch_x pitch_y frag_off seed_off koff_off pool_h pool_w skip "
            "col_off xtab_off small n_ks16 n_rt16 pp toep").split()
MM_RUN = ("kind zero_border in_img o_img o_origin o_row oc_pitch li_out expand x_img rec_per_img xtab_off pitch_x pitch_y sh ph pw n_ks n_rt "
          "frag_off seed_off koff_off col_off pix_per_img col_w out_c rs lo_clamp in_n small in_off o_off").split()
MM_HEAD = "ok batch waves buf_bytes x_bytes lds_bytes frag_lds frag_mode tbl_bytes frag_bytes n_seeds n_koff n_cols n_xtab n_intab pad_".split()
RUN_SKIP, RUN_MM, RUN_POOL4, RUN_POOL1, RUN_SOFTMAX = 0, 1, 2, 3, 4
POISON = 0x55


class Rec(dict):
    __getattr__ = dict.__getitem__


def _rec(names, words):
    assert len(names) == len(words), (len(names), len(words))
    return Rec(zip(names, (int(w) for w in words)))


class Plan:
    """The plans, fragments and seeds of an .ednn blob, fetched through edison_net_plan_dump."""

    def __init__(self, blob):
        from edison_amd import _lib
        L = _lib.lib()
        lay = [int(L.edison_net_plan_layout(i)) for i in range(12)]
        assert lay[2] == 4 * len(NET_LAYER) and lay[3] == 4 * len(MM_LAYER) and lay[4] == 4 * len(MM_RUN), "struct layouts changed: update tests/plan_emulator.py"
        buf = ctypes.create_string_buffer(bytes(blob), len(blob))
        fneed, sneed = ctypes.c_size_t(), ctypes.c_size_t()
        plan, mm = ctypes.create_string_buffer(lay[0]), ctypes.create_string_buffer(lay[1])
        r = L.edison_net_plan_dump(ctypes.cast(buf, ctypes.c_void_p), len(blob), plan, lay[0], mm, lay[1], None, 0, ctypes.byref(fneed), None, 0, ctypes.byref(sneed))
        if r != 0:
            raise _lib.EdisonError(r, "edison_net_plan_dump")
        frag, seeds = ctypes.create_string_buffer(max(fneed.value, 1)), ctypes.create_string_buffer(max(sneed.value, 4))
        r = L.edison_net_plan_dump(ctypes.cast(buf, ctypes.c_void_p), len(blob), None, 0, None, 0, frag, fneed.value, ctypes.byref(fneed), seeds, sneed.value, ctypes.byref(sneed))
        assert r == 0, r
        p32, m32 = np.frombuffer(plan.raw, np.int32), np.frombuffer(mm.raw, np.int32)
        self.P = _rec(NET_HEAD, p32[:len(NET_HEAD)])
        nl = self.P.n_layers
        self.PL = [_rec(NET_LAYER, p32[lay[5] // 4 + i * len(NET_LAYER):lay[5] // 4 + (i + 1) * len(NET_LAYER)]) for i in range(nl)]
        self.M = _rec(MM_HEAD, m32[:len(MM_HEAD)])
        self.ML = [_rec(MM_LAYER, m32[lay[6] // 4 + i * len(MM_LAYER):lay[6] // 4 + (i + 1) * len(MM_LAYER)]) for i in range(nl)]
        self.R = [_rec(MM_RUN, m32[lay[7] // 4 + i * len(MM_RUN):lay[7] // 4 + (i + 1) * len(MM_RUN)]) for i in range(nl)]
        self.koff = m32[lay[8] // 4:lay[9] // 4]
        self.coltab = m32[lay[9] // 4:lay[10] // 4]
        self.xtab = m32[lay[10] // 4:lay[11] // 4]
        self.intab = np.frombuffer(mm.raw, np.uint16, offset=lay[11])
        self.frag = np.frombuffer(frag.raw[:fneed.value], np.int8)
        self.seeds = np.frombuffer(seeds.raw[:sneed.value], np.int32)


class Slice:
    """One wave's LDS slice: the activation region and the expansion buffer behind it; every access is bounds-checked."""

    def __init__(self, n):
        self.b = np.full(n, POISON, np.uint8).view(np.int8)

    def rd(self, idx):
        idx = np.asarray(idx)
        assert idx.min() >= 0 and idx.max() < self.b.size, ("LDS read outside the slice", int(idx.min()), int(idx.max()), self.b.size)
        return self.b[idx]

    def wr(self, idx, v):
        idx = np.asarray(idx)
        assert idx.min() >= 0 and idx.max() < self.b.size, ("LDS write outside the slice", int(idx.min()), int(idx.max()), self.b.size)
        self.b[idx] = np.asarray(v, np.int64).astype(np.int8)


def _a_matrix(plan, R):
    """The layer's weight matrix [rows][chunks * 16] rebuilt from the operand fragments, in the kernel's chunk order."""
    if R.small:
        rows, per = 16 * R.n_rt, 4
    else:
        rows, per = 32 * R.n_rt, 2
    A = np.zeros((rows, per * R.n_ks * 16), np.int64)
    tile = rows // R.n_rt
    for rt in range(R.n_rt):
        for s in range(R.n_ks):
            f = plan.frag[R.frag_off + (rt * R.n_ks + s) * 1024:R.frag_off + (rt * R.n_ks + s + 1) * 1024].reshape(64, 16)
            for h in range(per):
                A[rt * tile:(rt + 1) * tile, (per * s + h) * 16:(per * s + h + 1) * 16] = f[h * tile:(h + 1) * tile]
    return A, per


def run(plan, x):
    """x [n][in_n] int8 -> dict(logits, softmax or None, argmax): `batch` inputs at a time through one LDS slice, as a wave does."""
    P, M = plan.P, plan.M
    x = np.ascontiguousarray(x, np.int8).reshape(-1, P.in_n)
    n, batch = x.shape[0], M.batch
    region = 2 * M.buf_bytes
    logits = np.zeros((n, P.out_n), np.int8)
    last = np.zeros((n, P.out_n), np.int8)
    for u0 in range(0, n, batch):
        nb = min(batch, n - u0)
        S = Slice(region + M.x_bytes)
        # ---- the inputs into layer 0's layout
        m0, in0 = plan.ML[0], plan.R[0].in_off
        if m0.in_hp != P.in_h or m0.in_wp != P.in_w:
            S.wr(in0 + np.arange(batch * m0.in_img), 0)
        for b in range(nb):
            if M.n_intab:
                at = plan.intab[:P.in_n].astype(np.int64)
            else:
                e = np.arange(P.in_n)
                pix, c = e // P.in_c, e % P.in_c
                at = ((pix // P.in_w + m0.in_py) * m0.in_wp + pix % P.in_w + m0.in_px) * P.in_c + c
            S.wr(in0 + b * m0.in_img + at, x[u0 + b])
        for li in range(P.n_layers):
            R = plan.R[li]
            if R.kind == RUN_SKIP:
                continue
            a, o = R.in_off, R.o_off
            assert a + batch * R.in_img <= region and o + batch * R.o_img <= region, "a layer's images leave the activation region"
            assert a + batch * R.in_img <= o or o + batch * R.o_img <= a, "a layer's input and output images overlap"
            if R.zero_border:
                S.wr(o + np.arange(batch * R.o_img), 0)
            if R.kind == RUN_MM:
                bsrc, img = a, R.in_img
                if R.expand:
                    L, ML = plan.PL[li], plan.ML[li]
                    dense = L.type == 3
                    out_w, in_c = (1, L.in_n) if dense else (L.out_w, L.in_c)
                    seg, sw = (1 if dense else L.kw) * in_c, 1 if dense else L.sw
                    for i in range(R.rec_per_img):
                        if R.xtab_off >= 0:
                            w0, doff = int(plan.xtab[2 * (R.xtab_off + i)]), int(plan.xtab[2 * (R.xtab_off + i) + 1])
                            soff, keep = w0 & 0xffffff, w0 >> 24
                        else:  # no table (ED_MM_MAX_XTAB): the kernel divides
                            r_, e2 = divmod(i, out_w * ML.cpr)
                            xo, j = divmod(e2, ML.cpr)
                            soff, keep = (r_ * ML.in_wp + xo * sw) * in_c + 16 * j, min(seg - 16 * j, 16)
                            doff = r_ * R.pitch_y + xo * R.pitch_x + 16 * j
                        for b in range(nb):
                            # the kernel's gather reads five aligned dwords around the 16 bytes
                            lo = (a + b * R.in_img + soff) & ~3
                            S.rd(np.arange(lo, lo + 20))
                            v = S.rd(a + b * R.in_img + soff + np.arange(16)).astype(np.int64)
                            v[keep:] = 0
                            S.wr(region + b * R.x_img + doff + np.arange(16), v)
                    bsrc, img = region, R.x_img
                A, per = _a_matrix(plan, R)
                nchunk = per * R.n_ks
                ko = plan.koff[R.koff_off:R.koff_off + nchunk].astype(np.int64)
                nwin = R.ph * R.pw
                rows = A.shape[0]
                seeds = plan.seeds[R.seed_off:R.seed_off + rows].astype(np.int64)
                for q in range(nb * R.pix_per_img):
                    b, pp = divmod(q, R.pix_per_img)
                    if R.col_off >= 0:
                        boff, ooff = int(plan.coltab[2 * (R.col_off + pp)]), int(plan.coltab[2 * (R.col_off + pp) + 1])
                    else:
                        y, xx = divmod(pp, R.col_w)
                        boff = (y * R.ph * R.sh) * R.pitch_y + (xx * R.pw) * R.pitch_x
                        ooff = R.o_origin + y * R.o_row + xx * R.oc_pitch
                    best = None
                    for w in range(nwin):
                        wy, wx = (w >> 1, w & 1) if R.pw == 2 else (w, 0)
                        base = bsrc + b * img + boff + (wy * R.sh) * R.pitch_y + wx * R.pitch_x
                        assert base % 16 == 0 and (ko % 16 == 0).all(), "a B fragment is not 16-byte aligned"
                        B = S.rd((base + ko)[:, None] + np.arange(16)[None, :]).astype(np.int64).reshape(-1)
                        acc = seeds + A @ B
                        best = acc if best is None else np.maximum(best, acc)
                    out = np.clip(best >> (R.rs & 0xff), R.lo_clamp, 127)[:R.out_c]
                    S.wr(o + b * R.o_img + ooff + np.arange(R.out_c), out)
            elif R.kind in (RUN_POOL4, RUN_POOL1):
                L = plan.PL[li]
                for b in range(nb):
                    for y in range(L.out_h):
                        for xx in range(L.out_w):
                            m = np.full(L.in_c, -129, np.int64)
                            for ky in range(L.kh):
                                iy = y * L.sh - L.pad_h + ky
                                if not 0 <= iy < L.in_h:
                                    continue
                                for kx in range(L.kw):
                                    ix = xx * L.sw - L.pad_w + kx
                                    if 0 <= ix < L.in_w:
                                        m = np.maximum(m, S.rd(a + b * R.in_img + (iy * L.in_w + ix) * L.in_c + np.arange(L.in_c)).astype(np.int64))
                            S.wr(o + b * R.o_img + R.o_origin + y * R.o_row + xx * R.oc_pitch + np.arange(L.in_c), m)
            else:  # softmax: arm_softmax_q7.c:215-260
                for b in range(nb):
                    v = S.rd(a + b * R.in_img + np.arange(R.in_n)).astype(np.int64)
                    base = int(v.max()) - 8
                    total = int((1 << np.clip(v - base, 0, 7)).sum())
                    ob = (1 << 20) // total
                    S.wr(o + b * R.o_img + np.arange(R.in_n), np.clip(ob >> np.clip(13 + base - v, 0, 31), -128, 127))
            for b in range(nb):
                if R.li_out == P.logits_layer:
                    logits[u0 + b] = S.rd(o + b * R.o_img + np.arange(P.out_n))
                if R.li_out == P.n_layers - 1:
                    last[u0 + b] = S.rd(o + b * R.o_img + np.arange(P.out_n))
    return dict(logits=logits, softmax=last if P.has_softmax else None, argmax=np.argmax(last.astype(np.int64), axis=1).astype(np.int32))
