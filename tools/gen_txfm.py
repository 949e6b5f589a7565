#!/usr/bin/env python3
"""Generate the 1-D transform butterfly networks for the AV1 path.

The AV1 inverse transforms are normative and defined as integer butterfly networks
(spec §7.13.2: B()/H() butterflies with Round2(.,12) after every rotation, cos128()/
sin128() constants = round(4096*cos(k*pi/128))).  This tool builds those networks
programmatically from the recursive structure of the inverse DCT (sizes 4..64) and the
explicit stage lists of the inverse ADST (4, 8, 16), checks each numerically against the
real-valued transform, and emits

  * av1-base_amd/csrc/txfm_gen.h   straight-line code (host+device) used by the HIP kernels
  * oracle/txfm_tables.h           op tables executed by the oracle's network interpreter

The FORWARD transforms (encoder side, non-normative) are the exact transposes of the inverse
networks (stages reversed, every 2x2 rotation transposed), so fwd(inv(x)) ~ (N/2) x.

Op encoding (one per output lane per stage):
  ('cp', a)               y = x[a]
  ('neg', a)              y = -x[a]
  ('add', a, b, sa, sb)   y = sa*x[a] + sb*x[b]          sa, sb in {+1,-1}
  ('rot', a, b, w0, w1)   y = Round2(w0*x[a] + w1*x[b], 12)
"""
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
COSPI = [int(round(4096 * math.cos(k * math.pi / 128))) for k in range(65)]
SINPI = [0, 1321, 2482, 3344, 3803]


def c(k):
    return COSPI[k]


def brev(v, bits):
    r = 0
    for i in range(bits):
        r = (r << 1) | ((v >> i) & 1)
    return r


def ident(n):
    return [('cp', i) for i in range(n)]


def idct_stages(n):
    """Stage list for the inverse DCT of size n = 2^k (k = 2..6)."""
    k = n.bit_length() - 1
    stages = []
    # stage 1: bit reversal
    stages.append([('cp', brev(i, k)) for i in range(n)])
    # collect per-size sub-networks: dct_sub[size] = list of stages over local indices,
    # applied to the slots [0, size) ; odd parts on [size/2, size)
    # Build as a list of "layers" where each layer is a dict slot->op, then align.
    def odd_part(N):
        """stages for slots N/2..N-1 of an N-point idct (after bit reversal)."""
        P = N // 2
        p = P.bit_length() - 1
        kk = N.bit_length() - 1
        u = 64 // N
        out = []
        # O1
        st = {}
        for j in range(P // 2):
            lo, hi = P + j, N - 1 - j
            m = brev(lo, kk)
            st[lo] = ('rot', lo, hi, c(64 - u * m), -c(u * m))
            st[hi] = ('rot', lo, hi, c(u * m), c(64 - u * m))
        out.append(st)
        for t in range(1, p):
            g = 1 << t
            st = {}
            for grp in range(P // g):
                base = P + grp * g
                for q in range(g // 2):
                    a, b = base + q, base + g - 1 - q
                    if grp % 2 == 0:
                        st[a] = ('add', a, b, 1, 1)
                        st[b] = ('add', a, b, 1, -1)
                    else:
                        st[a] = ('add', a, b, -1, 1)
                        st[b] = ('add', a, b, 1, 1)
            out.append(st)
            # rotation stage after as_t
            st = {}
            chunk = 1 << (t + 1)
            nch = max((P // 2) // chunk, 1)
            cb = nch.bit_length() - 1
            for ch in range(nch):
                a_ang = (128 // P) * (1 << (t - 1)) * (1 + 4 * brev(ch, cb)) if cb > 0 else (128 // P) * (1 << (t - 1))
                for q in range(chunk // 2):
                    lo_local = ch * chunk + chunk // 4 + q
                    if lo_local >= P // 2:
                        continue
                    lo, hi = P + lo_local, N - 1 - lo_local
                    typeA = q < chunk // 4
                    if typeA:
                        st[lo] = ('rot', lo, hi, -c(a_ang), c(64 - a_ang))
                        st[hi] = ('rot', lo, hi, c(64 - a_ang), c(a_ang))
                    else:
                        st[lo] = ('rot', lo, hi, -c(64 - a_ang), -c(a_ang))
                        st[hi] = ('rot', lo, hi, -c(a_ang), c(64 - a_ang))
            out.append(st)
        return out

    def final(N):
        st = {}
        for i in range(N // 2):
            st[i] = ('add', i, N - 1 - i, 1, 1)
            st[N - 1 - i] = ('add', i, N - 1 - i, 1, -1)
        return st

    def build(N):
        """list of sparse stages (dict slot->op) for idct N on slots [0,N) (post bit-reversal)."""
        if N == 2:
            return [{0: ('rot', 0, 1, c(32), c(32)), 1: ('rot', 0, 1, c(32), -c(32))}]
        ev = build(N // 2)
        od = odd_part(N)
        L = max(len(ev), len(od))
        # right-align both so they finish together
        ev = [{}] * (L - len(ev)) + ev
        od = [{}] * (L - len(od)) + od
        merged = []
        for a, b in zip(ev, od):
            d = dict(a)
            d.update(b)
            merged.append(d)
        merged.append(final(N))
        return merged

    for sp in build(n):
        stages.append([sp.get(i, ('cp', i)) for i in range(n)])
    return stages


def iadst4_ops():
    return None  # handled as special closed form (spec §7.13.2.6), see emit


def iadst8_stages():
    s = []
    s.append([('cp', i) for i in (7, 0, 5, 2, 3, 4, 1, 6)])
    st = []
    for j, (a, b) in enumerate(((4, 60), (20, 44), (36, 28), (52, 12))):
        st.append(('rot', 2 * j, 2 * j + 1, c(a), c(b)))
        st.append(('rot', 2 * j, 2 * j + 1, c(b), -c(a)))
    s.append(st)
    s.append([('add', i, i + 4, 1, 1) for i in range(4)] + [('add', i, i + 4, 1, -1) for i in range(4)])
    s.append([('cp', 0), ('cp', 1), ('cp', 2), ('cp', 3),
              ('rot', 4, 5, c(16), c(48)), ('rot', 4, 5, c(48), -c(16)),
              ('rot', 6, 7, -c(48), c(16)), ('rot', 6, 7, c(16), c(48))])
    s.append([('add', 0, 2, 1, 1), ('add', 1, 3, 1, 1), ('add', 0, 2, 1, -1), ('add', 1, 3, 1, -1),
              ('add', 4, 6, 1, 1), ('add', 5, 7, 1, 1), ('add', 4, 6, 1, -1), ('add', 5, 7, 1, -1)])
    s.append([('cp', 0), ('cp', 1), ('rot', 2, 3, c(32), c(32)), ('rot', 2, 3, c(32), -c(32)),
              ('cp', 4), ('cp', 5), ('rot', 6, 7, c(32), c(32)), ('rot', 6, 7, c(32), -c(32))])
    s.append([('cp', 0), ('neg', 4), ('cp', 6), ('neg', 2), ('cp', 3), ('neg', 7), ('cp', 5), ('neg', 1)])
    return s


def iadst16_stages():
    s = []
    s.append([('cp', i) for i in (15, 0, 13, 2, 11, 4, 9, 6, 7, 8, 5, 10, 3, 12, 1, 14)])
    st = []
    for j, (a, b) in enumerate(((2, 62), (10, 54), (18, 46), (26, 38), (34, 30), (42, 22), (50, 14), (58, 6))):
        st.append(('rot', 2 * j, 2 * j + 1, c(a), c(b)))
        st.append(('rot', 2 * j, 2 * j + 1, c(b), -c(a)))
    s.append(st)
    s.append([('add', i, i + 8, 1, 1) for i in range(8)] + [('add', i, i + 8, 1, -1) for i in range(8)])
    s.append([('cp', i) for i in range(8)] + [
        ('rot', 8, 9, c(8), c(56)), ('rot', 8, 9, c(56), -c(8)),
        ('rot', 10, 11, c(40), c(24)), ('rot', 10, 11, c(24), -c(40)),
        ('rot', 12, 13, -c(56), c(8)), ('rot', 12, 13, c(8), c(56)),
        ('rot', 14, 15, -c(24), c(40)), ('rot', 14, 15, c(40), c(24))])
    s.append([('add', 0, 4, 1, 1), ('add', 1, 5, 1, 1), ('add', 2, 6, 1, 1), ('add', 3, 7, 1, 1),
              ('add', 0, 4, 1, -1), ('add', 1, 5, 1, -1), ('add', 2, 6, 1, -1), ('add', 3, 7, 1, -1),
              ('add', 8, 12, 1, 1), ('add', 9, 13, 1, 1), ('add', 10, 14, 1, 1), ('add', 11, 15, 1, 1),
              ('add', 8, 12, 1, -1), ('add', 9, 13, 1, -1), ('add', 10, 14, 1, -1), ('add', 11, 15, 1, -1)])
    s.append([('cp', 0), ('cp', 1), ('cp', 2), ('cp', 3),
              ('rot', 4, 5, c(16), c(48)), ('rot', 4, 5, c(48), -c(16)),
              ('rot', 6, 7, -c(48), c(16)), ('rot', 6, 7, c(16), c(48)),
              ('cp', 8), ('cp', 9), ('cp', 10), ('cp', 11),
              ('rot', 12, 13, c(16), c(48)), ('rot', 12, 13, c(48), -c(16)),
              ('rot', 14, 15, -c(48), c(16)), ('rot', 14, 15, c(16), c(48))])
    st = []
    for b in (0, 4, 8, 12):
        st += [('add', b, b + 2, 1, 1), ('add', b + 1, b + 3, 1, 1), ('add', b, b + 2, 1, -1), ('add', b + 1, b + 3, 1, -1)]
    s.append(st)
    st = []
    for b in (0, 4, 8, 12):
        st += [('cp', b), ('cp', b + 1), ('rot', b + 2, b + 3, c(32), c(32)), ('rot', b + 2, b + 3, c(32), -c(32))]
    s.append(st)
    s.append([('cp', 0), ('neg', 8), ('cp', 12), ('neg', 4), ('cp', 6), ('neg', 14), ('cp', 10), ('neg', 2),
              ('cp', 3), ('neg', 11), ('cp', 15), ('neg', 7), ('cp', 5), ('neg', 13), ('cp', 9), ('neg', 1)])
    return s


# ------------------------------------------------------------------------- evaluation
def round2(v, n):
    return (v + (1 << (n - 1))) >> n


def run(stages, x):
    x = [int(v) for v in x]
    for st in stages:
        y = []
        for op in st:
            if op[0] == 'cp':
                y.append(x[op[1]])
            elif op[0] == 'neg':
                y.append(-x[op[1]])
            elif op[0] == 'add':
                y.append(op[3] * x[op[1]] + op[4] * x[op[2]])
            else:
                y.append(round2(op[3] * x[op[1]] + op[4] * x[op[2]], 12))
        x = y
    return x


def transpose(stages):
    """Exact transpose network: y = M x  ->  x' = M^T y (with the same Round2 per rotation)."""
    n = len(stages[0])
    out = []
    for st in reversed(stages):
        contrib = [[] for _ in range(n)]  # contrib[src] = list of (dst, weight, is_rot)
        for dst, op in enumerate(st):
            if op[0] == 'cp':
                contrib[op[1]].append((dst, 1, False))
            elif op[0] == 'neg':
                contrib[op[1]].append((dst, -1, False))
            elif op[0] == 'add':
                contrib[op[1]].append((dst, op[3], False))
                contrib[op[2]].append((dst, op[4], False))
            else:
                contrib[op[1]].append((dst, op[3], True))
                contrib[op[2]].append((dst, op[4], True))
        new = []
        for src in range(n):
            cs = contrib[src]
            if len(cs) == 1 and not cs[0][2]:
                new.append(('cp', cs[0][0]) if cs[0][1] == 1 else ('neg', cs[0][0]))
            elif len(cs) == 2 and not cs[0][2] and not cs[1][2]:
                new.append(('add', cs[0][0], cs[1][0], cs[0][1], cs[1][1]))
            elif len(cs) == 2 and cs[0][2] and cs[1][2]:
                new.append(('rot', cs[0][0], cs[1][0], cs[0][1], cs[1][1]))
            else:
                raise AssertionError(("untransposable", src, cs))
        out.append(new)
    return out


def check():
    rng = np.random.default_rng(1)
    for n in (4, 8, 16, 32, 64):
        st = idct_stages(n)
        X = rng.integers(-2000, 2000, n)
        got = np.array(run(st, X), dtype=float)
        nn = np.arange(n)
        ref = np.array([X[0] / math.sqrt(2) + sum(X[k] * math.cos((2 * m + 1) * k * math.pi / (2 * n)) for k in range(1, n)) for m in nn])
        err = np.abs(got - ref).max()
        assert err < (8.0 if n == 64 else 4.0), (n, err)
        # forward = transpose; fwd(inv(X)) ~ (n/2) X
        ft = transpose(st)
        back = np.array(run(ft, run(st, X)), dtype=float)
        e2 = np.abs(back / (n / 2) - X).max()
        assert e2 < 3.0, (n, e2)
        print("idct%-2d ok: max err vs real transform %.2f, fwd(inv) err %.2f, stages %d" % (n, err, e2, len(st)))
    for n, st in ((8, iadst8_stages()), (16, iadst16_stages())):
        X = rng.integers(-2000, 2000, n)
        got = np.array(run(st, X), dtype=float)
        # AV1 ADST (n=8,16): x[m] = sum_k X[k] sin(pi (2m+1)(2k+1) / (4n))
        ref = np.array([sum(X[k] * math.sin(math.pi * (2 * m + 1) * (2 * k + 1) / (4 * n)) for k in range(n)) for m in range(n)])
        err = np.abs(got - ref).max()
        assert err < 4.0, ("iadst", n, err, got[:4], ref[:4])
        ft = transpose(st)
        back = np.array(run(ft, run(st, X)), dtype=float)
        e2 = np.abs(back / (n / 2) - X).max()
        assert e2 < 3.0, (n, e2)
        print("iadst%-2d ok: max err %.2f, fwd(inv) err %.2f" % (n, err, e2))


# ------------------------------------------------------------------------- emission
def emit_straightline(name, stages, lines):
    n = len(stages[0])
    lines.append("AV1_TXFM_FN void %s(int32_t *x) {" % name)
    cur = ["x[%d]" % i for i in range(n)]
    tmp_id = 0
    for si, st in enumerate(stages):
        nxt = [None] * n
        decl = []
        for dst, op in enumerate(st):
            if op[0] == 'cp':
                nxt[dst] = cur[op[1]]
                continue
            v = "t%d" % tmp_id
            tmp_id += 1
            if op[0] == 'neg':
                decl.append("const int32_t %s = -%s;" % (v, cur[op[1]]))
            elif op[0] == 'add':
                sa = "" if op[3] == 1 else "-"
                sb = "+" if op[4] == 1 else "-"
                decl.append("const int32_t %s = %s%s %s %s;" % (v, sa, cur[op[1]], sb, cur[op[2]]))
            else:
                decl.append("const int32_t %s = av1_half_btf(%d, %s, %d, %s);" % (v, op[3], cur[op[1]], op[4], cur[op[2]]))
            nxt[dst] = v
        lines.append("  /* stage %d */" % (si + 1))
        for d_ in decl:
            lines.append("  " + d_)
        cur = nxt
    # final write back (values may alias x[]: go through temporaries)
    for i in range(n):
        if cur[i].startswith("x["):
            lines.append("  const int32_t o%d = %s;" % (i, cur[i]))
            cur[i] = "o%d" % i
    for i in range(n):
        lines.append("  x[%d] = %s;" % (i, cur[i]))
    lines.append("}")
    lines.append("")


def emit_pruned(name, stages, nz, lines):
    """The same network with x[nz ..] known to be zero (the inverse transform of a block whose coefficients beyond row / column nz are
    all zero: most blocks): zeros are propagated symbolically - a rotation with one zero input is ONE multiply (Round2(w * a, 12), exactly
    what the full form computes), with two it is zero; sums with a zero are copies.  Reads x[0 .. nz), writes all n outputs."""
    n = len(stages[0])
    lines.append("AV1_TXFM_FN void %s_nz%d(int32_t *x) {" % (name, nz))
    cur = [("x[%d]" % i) if i < nz else None for i in range(n)]   # None = zero
    tmp_id = 0
    for si, st in enumerate(stages):
        nxt = [None] * n
        decl = []
        for dst, op in enumerate(st):
            if op[0] == 'cp':
                nxt[dst] = cur[op[1]]
                continue
            if op[0] == 'neg':
                if cur[op[1]] is None:
                    continue
                e = "-%s" % cur[op[1]]
            elif op[0] == 'add':
                a, b = cur[op[1]], cur[op[2]]
                if a is None and b is None:
                    continue
                if b is None:
                    if op[3] == 1:
                        nxt[dst] = a
                        continue
                    e = "-%s" % a
                elif a is None:
                    if op[4] == 1:
                        nxt[dst] = b
                        continue
                    e = "-%s" % b
                else:
                    e = "%s%s %s %s" % ("" if op[3] == 1 else "-", a, "+" if op[4] == 1 else "-", b)
            else:
                a, b = cur[op[1]], cur[op[2]]
                if a is None and b is None:
                    continue
                if b is None:
                    e = "av1_half_btf1(%d, %s)" % (op[3], a)
                elif a is None:
                    e = "av1_half_btf1(%d, %s)" % (op[4], b)
                else:
                    e = "av1_half_btf(%d, %s, %d, %s)" % (op[3], a, op[4], b)
            v = "t%d" % tmp_id
            tmp_id += 1
            decl.append("const int32_t %s = %s;" % (v, e))
            nxt[dst] = v
        if decl:
            lines.append("  /* stage %d */" % (si + 1))
            for d_ in decl:
                lines.append("  " + d_)
        cur = nxt
    for i in range(n):
        if cur[i] is not None and cur[i].startswith("x["):
            lines.append("  const int32_t o%d = %s;" % (i, cur[i]))
            cur[i] = "o%d" % i
    for i in range(n):
        lines.append("  x[%d] = %s;" % (i, cur[i] if cur[i] is not None else "0"))
    lines.append("}")
    lines.append("")


def run_pruned(stages, x, nz):
    """executes what emit_pruned emits (same zero propagation), for check_pruned"""
    n = len(stages[0])
    cur = [int(v) if i < nz else None for i, v in enumerate(x)]
    for st in stages:
        nxt = [None] * n
        for dst, op in enumerate(st):
            if op[0] == 'cp':
                nxt[dst] = cur[op[1]]
            elif op[0] == 'neg':
                nxt[dst] = None if cur[op[1]] is None else -cur[op[1]]
            elif op[0] == 'add':
                a, b = cur[op[1]], cur[op[2]]
                if a is None and b is None:
                    nxt[dst] = None
                else:
                    nxt[dst] = op[3] * (a or 0) + op[4] * (b or 0)
            else:
                a, b = cur[op[1]], cur[op[2]]
                if a is None and b is None:
                    nxt[dst] = None
                elif b is None:
                    nxt[dst] = round2(op[3] * a, 12)
                elif a is None:
                    nxt[dst] = round2(op[4] * b, 12)
                else:
                    nxt[dst] = round2(op[3] * a + op[4] * b, 12)
        cur = nxt
    return [0 if v is None else v for v in cur]


# inverse networks that get pruned forms, and the input counts (the kernels pick the smallest that covers a block's nonzero extent)
PRUNED = {"av1_idct8": (4,), "av1_iadst8": (4,), "av1_idct16": (4, 8), "av1_iadst16": (4, 8), "av1_idct32": (4, 8, 16), "av1_idct64": (8, 16)}


def check_pruned(nets):
    rng = np.random.default_rng(2)
    for name, nzs in PRUNED.items():
        st = nets[name]
        n = len(st[0])
        for nz in nzs:
            for _ in range(200):
                x = [int(v) for v in rng.integers(-(1 << 17), 1 << 17, n)]
                x[nz:] = [0] * (n - nz)
                assert run_pruned(st, x, nz) == run(st, x), (name, nz)
    print("pruned networks == full networks on zero-extended inputs")


def emit_tables(name, stages, lines):
    """int16 ops: kind, a, b, w0, w1 per lane per stage. kind: 0 cp 1 neg 2 add 3 rot"""
    n = len(stages[0])
    flat = []
    for st in stages:
        for op in st:
            if op[0] == 'cp':
                flat += [0, op[1], 0, 0, 0]
            elif op[0] == 'neg':
                flat += [1, op[1], 0, 0, 0]
            elif op[0] == 'add':
                flat += [2, op[1], op[2], op[3], op[4]]
            else:
                flat += [3, op[1], op[2], op[3], op[4]]
    lines.append("static const int16_t %s_ops[%d * %d * 5] = {" % (name, len(stages), n))
    for i in range(0, len(flat), 20):
        lines.append("  " + ", ".join(str(v) for v in flat[i:i + 20]) + ",")
    lines.append("};")
    lines.append("#define %s_NSTAGES %d" % (name.upper(), len(stages)))
    lines.append("")


def main():
    check()
    nets = {}
    for n in (4, 8, 16, 32, 64):
        st = idct_stages(n)
        nets["av1_idct%d" % n] = st
        nets["av1_fdct%d" % n] = transpose(st)
    for n, st in ((8, iadst8_stages()), (16, iadst16_stages())):
        nets["av1_iadst%d" % n] = st
        nets["av1_fadst%d" % n] = transpose(st)

    hdr = ["/* GENERATED by tools/gen_txfm.py - do not edit.",
           " * Straight-line integer butterfly networks: normative AV1 inverse DCT/ADST (spec §7.13.2)",
           " * and their exact transposes as forward transforms.  Each function transforms x[0..n) in",
           " * place.  av1_half_btf(w0,a,w1,b) = Round2(w0*a + w1*b, 12) with 4096*cos(k*pi/128) weights. */",
           "#ifndef AV1MI_TXFM_GEN_H", "#define AV1MI_TXFM_GEN_H", "#include <stdint.h>",
           "#ifndef AV1_TXFM_FN", "#define AV1_TXFM_FN static inline", "#endif",
           "#ifndef AV1_HALF_BTF_DEFINED  /* an includer may bring its own (the HIP kernels: 24-bit multiplies) */",
           "AV1_TXFM_FN int32_t av1_half_btf(int32_t w0, int32_t a, int32_t w1, int32_t b) {",
           "  return (int32_t)(((int64_t)w0 * a + (int64_t)w1 * b + 2048) >> 12);", "}", "#endif", ""]
    hdr += ["/* Round2(w * a, 12): a rotation whose other input is known to be zero (the pruned networks below) */",
            "#ifndef AV1_HALF_BTF1_DEFINED",
            "AV1_TXFM_FN int32_t av1_half_btf1(int32_t w, int32_t a) { return (int32_t)(((int64_t)w * a + 2048) >> 12); }", "#endif", ""]
    for name, st in nets.items():
        emit_straightline(name, st, hdr)
    check_pruned(nets)
    for name, nzs in PRUNED.items():
        for nz in nzs:
            emit_pruned(name, nets[name], nz, hdr)
    hdr.append("#endif")
    p1 = os.path.join(HERE, "..", "av1-base_amd", "csrc", "txfm_gen.h")
    open(p1, "w").write("\n".join(hdr) + "\n")

    tb = ["/* GENERATED by tools/gen_txfm.py - do not edit.  Op tables for the oracle's network",
          " * interpreter (oracle/av1o_txfm.c): 5 int16 per lane per stage = kind,a,b,w0,w1;",
          " * kind 0 copy x[a], 1 negate x[a], 2 w0*x[a]+w1*x[b] (w=+-1), 3 Round2(w0*x[a]+w1*x[b],12). */",
          "#ifndef AV1O_TXFM_TABLES_H", "#define AV1O_TXFM_TABLES_H", "#include <stdint.h>", ""]
    for name, st in nets.items():
        emit_tables(name, st, tb)
    tb.append("#endif")
    p2 = os.path.join(HERE, "..", "oracle", "txfm_tables.h")
    open(p2, "w").write("\n".join(tb) + "\n")
    print("wrote", os.path.relpath(p1), "and", os.path.relpath(p2))
    emit_fdct32_matrix()


def fdct32_matrix():
    """Cm[k][n] = round(4096 * orthonormal 32-point DCT-II): the luma 32x32 forward transform as an exact-integer matrix product
    (the matrix cores run it: v_mfma_i32_32x32x32_i8 on signed-byte halves of matrix and data)."""
    return [[int(round(4096.0 * math.sqrt(2.0 / 32) * (math.sqrt(0.5) if k == 0 else 1.0) * math.cos((2 * n + 1) * k * math.pi / 64)))
             for n in range(32)] for k in range(32)]


def emit_fdct32_matrix():
    cm = fdct32_matrix()
    assert max(abs(v) for r in cm for v in r) <= 1024
    lo8 = lambda v: v & 255
    hi8 = lambda v: ((v + 128) >> 8) & 255
    pack = lambda b: b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24)
    out = ["/* GENERATED by tools/gen_txfm.py - do not edit.  The luma 32x32 forward transform as a matrix product:",
           " *   U[r][m] = (sum_c X[r][c] * Cm[m][c] + 512) >> 10,   Y[k][m] = (sum_r Cm[k][r] * U[r][m] + 2048) >> 12",
           " * with Cm = round(4096 * orthonormal DCT-II) - 4 x the orthonormal 2-D DCT, the scale of the butterfly form.",
           " * AV1_FDCT32_FRAG: per lane of a wave the four operand fragments of v_mfma_i32_32x32x32_i8 (16 signed bytes each,",
           " * value = 256 * hi + lo, lo = (int8)(v & 255), hi = (v + 128) >> 8): stage-1 B = Cm^T low, high (lane = column m, bytes",
           " * = Cm[m][16h + j]); stage-2 A = Cm low, high with k in the order the accumulator registers hold rows",
           " * (byte j of lane half h = Cm[k][(j & 3) + 8 * (j >> 2) + 4 * h]). */",
           "#ifndef AV1MI_FDCT32_MATRIX_H", "#define AV1MI_FDCT32_MATRIX_H", "#include <stdint.h>",
           "#define AV1_FDCT32_MATRIX_INIT { \\"]
    for r in cm:
        out.append("  { " + ", ".join("%d" % v for v in r) + " }, \\")
    out.append("}")
    out.append("#define AV1_FDCT32_FRAG_INIT { \\")
    for lane in range(64):
        r, h = lane & 31, lane >> 5
        s1 = [cm[r][16 * h + j] for j in range(16)]
        s2 = [cm[r][(j & 3) + 8 * (j >> 2) + 4 * h] for j in range(16)]
        words = []
        for vals, f in ((s1, lo8), (s1, hi8), (s2, lo8), (s2, hi8)):
            b = [f(v) for v in vals]
            words += [pack(b[4 * i:4 * i + 4]) for i in range(4)]
        out.append("  { " + ", ".join("0x%08xu" % w for w in words) + " }, \\")
    out.append("}")
    out.append("#endif")
    for path in (os.path.join(HERE, "..", "av1-base_amd", "csrc", "fdct32_matrix.h"), os.path.join(HERE, "..", "oracle", "fdct32_matrix.h")):
        open(path, "w").write("\n".join(out) + "\n")
        print("wrote", os.path.relpath(path))


if __name__ == "__main__":
    main()
