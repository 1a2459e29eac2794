#!/usr/bin/env python3
"""Generates tools/aes_bitslice_gen.inc: the AES S-box as a circuit of three-input look-up operations (gfx950's
v_bitop3_b32: any Boolean function of three 32-bit words, one instruction) for the bit-sliced AES microbenchmark
tools/aes_bitslice_bench.hip.

Source circuit: Boyar & Peralta's depth-16 S-box (128 two-input gates: 32 AND, 92 XOR, 4 XNOR; "A depth-16 circuit for the AES
S-box", 2011), restated below and checked here against the S-box computed from its definition (inverse in GF(2^8) mod
x^8+x^4+x^3+x+1, then the affine map) on all 256 inputs.  Mapping: every k <= 3 feasible cut of every gate is enumerated with its
truth table, then a cover is chosen by area flow, exact-area refinement passes (the FPGA technology-mapping recipe: 90
operations) and simulated annealing over the cut choices (82).  Flattening the linear layers and re-extracting common
three-way xors greedily was tried and is worse (107): the source circuit's sharing is already good.  The result is verified again on all 256 inputs before anything is written.

usage: gen_aes_bitslice.py [--stats]   (writes tools/aes_bitslice_gen.inc)"""
import itertools
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))

BP = """
T1 = U0 + U3;T2 = U0 + U5;T3 = U0 + U6;T4 = U3 + U5;T5 = U4 + U6;T6 = T1 + T5;T7 = U1 + U2;T8 = U7 + T6;T9 = U7 + T7
T10 = T6 + T7;T11 = U1 + U5;T12 = U2 + U5;T13 = T3 + T4;T14 = T6 + T11;T15 = T5 + T11;T16 = T5 + T12;T17 = T9 + T16
T18 = U3 + U7;T19 = T7 + T18;T20 = T1 + T19;T21 = U6 + U7;T22 = T7 + T21;T23 = T2 + T22;T24 = T2 + T10;T25 = T20 + T17
T26 = T3 + T16;T27 = T1 + T12
M1 = T13 x T6;M2 = T23 x T8;M3 = T14 + M1;M4 = T19 x U7;M5 = M4 + M1;M6 = T3 x T16;M7 = T22 x T9;M8 = T26 + M6
M9 = T20 x T17;M10 = M9 + M6;M11 = T1 x T15;M12 = T4 x T27;M13 = M12 + M11;M14 = T2 x T10;M15 = M14 + M11;M16 = M3 + M2
M17 = M5 + T24;M18 = M8 + M7;M19 = M10 + M15;M20 = M16 + M13;M21 = M17 + M15;M22 = M18 + M13;M23 = M19 + T25
M24 = M22 + M23;M25 = M22 x M20;M26 = M21 + M25;M27 = M20 + M21;M28 = M23 + M25;M29 = M28 x M27;M30 = M26 x M24
M31 = M20 x M23;M32 = M27 x M31;M33 = M27 + M25;M34 = M21 x M22;M35 = M24 x M34;M36 = M24 + M25;M37 = M21 + M29
M38 = M32 + M33;M39 = M23 + M30;M40 = M35 + M36;M41 = M38 + M40;M42 = M37 + M39;M43 = M37 + M38;M44 = M39 + M40
M45 = M42 + M41;M46 = M44 x T6;M47 = M40 x T8;M48 = M39 x U7;M49 = M43 x T16;M50 = M38 x T9;M51 = M37 x T17
M52 = M42 x T15;M53 = M45 x T27;M54 = M41 x T10;M55 = M44 x T13;M56 = M40 x T23;M57 = M39 x T19;M58 = M43 x T3
M59 = M38 x T22;M60 = M37 x T20;M61 = M42 x T1;M62 = M45 x T4;M63 = M41 x T2
L0 = M61 + M62;L1 = M50 + M56;L2 = M46 + M48;L3 = M47 + M55;L4 = M54 + M58;L5 = M49 + M61;L6 = M62 + L5;L7 = M46 + L3
L8 = M51 + M59;L9 = M52 + M53;L10 = M53 + L4;L11 = M60 + L2;L12 = M48 + M51;L13 = M50 + L0;L14 = M52 + M61
L15 = M55 + L1;L16 = M56 + L0;L17 = M57 + L1;L18 = M58 + L8;L19 = M63 + L4;L20 = L0 + L1;L21 = L1 + L7;L22 = L3 + L12
L23 = L18 + L2;L24 = L15 + L9;L25 = L6 + L10;L26 = L7 + L9;L27 = L8 + L10;L28 = L11 + L14;L29 = L11 + L17
S0 = L6 + L24;S1 = L16 # L26;S2 = L19 # L28;S3 = L6 + L21;S4 = L20 + L22;S5 = L25 + L29;S6 = L13 # L27;S7 = L6 # L23
"""
ANNEAL_SEEDS = 6
INPUTS = [f"U{i}" for i in range(8)]     # U0 = most significant bit of the input byte
OUTPUTS = [f"S{i}" for i in range(8)]    # S0 = most significant bit of the output byte


def sbox_table():
    def mul(a, b):
        r = 0
        while b:
            if b & 1:
                r ^= a
            a = (a << 1) ^ (0x11B if a & 0x80 else 0)
            b >>= 1
        return r & 0xFF
    inv = [0] * 256
    for x in range(1, 256):
        inv[x] = next(y for y in range(1, 256) if mul(x, y) == 1)
    out = []
    for x in range(256):
        v = r = inv[x]
        for _ in range(4):
            v = ((v << 1) | (v >> 7)) & 0xFF
            r ^= v
        out.append(r ^ 0x63)
    return out


def parse():
    gates = {}
    order = []
    for stmt in BP.replace("\n", ";").split(";"):
        stmt = stmt.strip()
        if not stmt:
            continue
        d, _, a, op, b = stmt.split()
        gates[d] = (op, a, b)
        order.append(d)
    return gates, order


def column(gates, order):
    """value of every signal as a 256-bit integer: bit x = the signal on input byte x"""
    val = {}
    for i, u in enumerate(INPUTS):
        val[u] = sum(((x >> (7 - i)) & 1) << x for x in range(256))
    full = (1 << 256) - 1
    for d in order:
        op, a, b = gates[d]
        val[d] = val[a] ^ val[b] if op == "+" else val[a] & val[b] if op == "x" else full ^ val[a] ^ val[b]
    return val


def truth_table(val, node, leaves):
    """8-bit table of `node` as a function of up to three leaves (table bit = 4 a + 2 b + c, v_bitop3's convention with the
    leaves as operands a, b, c); None if node is not a function of the leaves alone"""
    ls = list(leaves) + [None] * (3 - len(leaves))
    tt, seen = 0, {}
    for x in range(256):
        key = tuple(((val[l] >> x) & 1) if l else 0 for l in ls)
        bit = (val[node] >> x) & 1
        if seen.setdefault(key, bit) != bit:
            return None
    for key, bit in seen.items():
        idx = 4 * key[0] + 2 * key[1] + key[2]
        tt |= bit << idx
    # unreachable leaf combinations keep 0; fill the don't-care positions of absent operands so the table ignores them
    for idx in range(8):
        key = ((idx >> 2) & 1, (idx >> 1) & 1, idx & 1)
        base = tuple(k if l else 0 for k, l in zip(key, ls))
        if base in seen and seen[base]:
            tt |= 1 << idx
    return tt


def cuts_of(gates, order):
    cuts = {u: [frozenset([u])] for u in INPUTS}
    for d in order:
        _, a, b = gates[d]
        cs = {frozenset([d])}
        for ca in cuts[a]:
            for cb in cuts[b]:
                c = ca | cb
                if len(c) <= 3:
                    cs.add(c)
        cuts[d] = sorted(cs, key=lambda c: (len(c), sorted(c)))
    return cuts


def map_lut3(gates, order, val, passes=8):
    cuts = cuts_of(gates, order)
    fanout = {s: 0 for s in list(INPUTS) + order}
    for d in order:
        fanout[gates[d][1]] += 1
        fanout[gates[d][2]] += 1
    for o in OUTPUTS:
        fanout[o] += 1
    # area flow
    best, flow = {}, {u: 0.0 for u in INPUTS}
    for d in order:
        cand = []
        for c in cuts[d]:
            if c == frozenset([d]):
                continue
            cand.append((1 + sum(flow[l] / max(1, fanout[l]) for l in c), sorted(c)))
        f, c = min(cand)
        best[d], flow[d] = frozenset(c), f

    def cover(choice):
        need, stack = set(), list(OUTPUTS)
        while stack:
            n = stack.pop()
            if n in need or n in INPUTS:
                continue
            need.add(n)
            stack.extend(choice[n])
        return need

    cur = cover(best)
    # exact-area refinement: re-choose each used node's cut to minimise the size of the whole cover
    for _ in range(passes):
        improved = False
        for d in reversed(order):
            if d not in cur:
                continue
            keep = best[d]
            best_size, best_cut = len(cur), keep
            for c in cuts[d]:
                if c == frozenset([d]) or c == keep:
                    continue
                best[d] = c
                size = len(cover(best))
                if size < best_size:
                    best_size, best_cut = size, c
            best[d] = best_cut
            if best_cut != keep:
                improved = True
                cur = cover(best)
        if not improved:
            break
    # annealing over the cut choices (cost = size of the cover): the greedy passes end at 90 operations, this at 82
    import math
    import random
    cand = {d: [c for c in cuts[d] if c != frozenset([d])] for d in order}
    overall = (len(cur), dict(best))
    for seed in range(ANNEAL_SEEDS):
        rnd = random.Random(seed)
        choice = dict(overall[1])
        size, temp = len(cover(choice)), 1.0
        for _ in range(60000):
            d = rnd.choice(order)
            if len(cand[d]) < 2:
                continue
            old = choice[d]
            choice[d] = rnd.choice(cand[d])
            ns = len(cover(choice))
            if ns <= size or rnd.random() < math.exp((size - ns) / temp):
                size = ns
                if size < overall[0]:
                    overall = (size, dict(choice))
            else:
                choice[d] = old
            temp = max(0.05, temp * 0.9999)
    best = overall[1]
    cur = cover(best)
    used = [d for d in order if d in cur]
    ops = []
    for d in used:
        leaves = sorted(best[d], key=lambda s: (s not in INPUTS, order.index(s) if s in order else -1))
        tt = truth_table(val, d, leaves)
        assert tt is not None, (d, leaves)
        ops.append((d, leaves, tt))
    return ops


def verify(ops):
    S = sbox_table()
    for x in range(256):
        v = {u: (x >> (7 - i)) & 1 for i, u in enumerate(INPUTS)}
        for d, leaves, tt in ops:
            a, b, c = ([v[l] for l in leaves] + [0, 0])[:3]
            v[d] = (tt >> (4 * a + 2 * b + c)) & 1
        got = sum(v[o] << (7 - i) for i, o in enumerate(OUTPUTS))
        assert got == S[x], (x, got, S[x])


def main():
    gates, order = parse()
    val = column(gates, order)
    S = sbox_table()
    for x in range(256):
        assert sum(((val[o] >> x) & 1) << (7 - i) for i, o in enumerate(OUTPUTS)) == S[x]
    ops = map_lut3(gates, order, val)
    verify(ops)
    kinds = {}
    for _, leaves, tt in ops:
        k = "xor3" if tt == 0x96 else "xor2" if len(leaves) == 2 and tt in (0x3C, 0x66, 0x5A) else f"{len(leaves)}-input"
        kinds[k] = kinds.get(k, 0) + 1
    if "--stats" in sys.argv:
        print(f"two-input gates {len(order)} -> three-input operations {len(ops)}: {kinds}")
    lines = ["// GENERATED by tools/gen_aes_bitslice.py -- do not edit.",
             f"// The AES S-box on bit planes: {len(ops)} three-input operations (v_bitop3_b32) mapped from Boyar & Peralta's depth-16",
             f"// circuit of {len(order)} two-input gates; checked against the S-box on all 256 inputs by the generator.",
             "// u[0] / s[0] = the most significant bit plane of the byte.  LUT3(table, a, b, c): table bit 4a + 2b + c.",
             "template <class W>", "__device__ __forceinline__ void aes_sbox_planes(W (&s)[8], const W (&u)[8]) {"]
    name = {f"U{i}": f"u[{i}]" for i in range(8)}
    for d, leaves, tt in ops:
        args = [name[l] for l in leaves] + ["W(0)"] * (3 - len(leaves))
        name[d] = d.lower()
        lines.append(f"  const W {d.lower()} = LUT3(0x{tt:02x}, {', '.join(args)});")
    for i, o in enumerate(OUTPUTS):
        lines.append(f"  s[{i}] = {name[o]};")
    lines.append("}")
    lines.append(f"constexpr int AES_SBOX_LUT3_OPS = {len(ops)};")
    with open(os.path.join(HERE, "aes_bitslice_gen.inc"), "w") as fh:
        fh.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
