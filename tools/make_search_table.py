#!/usr/bin/env python3
"""Generate limg_amd/csrc/limg_search_table.h: the decision automaton of the reference's default ("fast") shift search.

The search (guess src/limg_bit_crush.h:331-392, then stepwise coarse + fine :502-614) is a deterministic program whose only
inputs are the pass / fail outcomes of the trials it asks for.  Its control state space is small, so the whole program
can be unrolled into a table: state -> (shift triple to try, next state on pass, next state on fail), with terminal states
carrying the resulting shift.  The kernel then runs ONE trial loop driven by scalar loads from this table instead of six
inlined copies of the trial inside nested uniform loops.

`search_fast` below is a literal restatement of the upstream control flow as a generator (same loop-carried resets,
uint8 counters never exceed 10 here); the table is produced by exhaustively exploring both outcomes of every trial and
merging identical control states.  tests/test_host.py replays it against the golden trial tables of the real reference.
"""
import os
import sys

sys.setrecursionlimit(100000)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def search_fast():
    shift = [0, 0, 0]
    ok = yield (4, 5, 6)
    if ok:
        shift = [4, 5, 6]
        ok = yield (5, 8, 8)
        if ok:
            shift = [5, 8, 8]
        else:
            ok = yield (4, 6, 8)
            if ok:
                shift = [4, 6, 8]
    else:
        ok = yield (2, 4, 5)
        if ok:
            shift = [2, 4, 5]
    max_shift = sum(shift)
    a = shift[0] & 15
    b = shift[1] & 15
    c = (shift[2] & 15) + 2
    while a <= 8:
        while b <= 8:
            while c <= 8:
                if a + b + c > max_shift:
                    ok = yield (a, b, c)
                    if ok:
                        shift = [a, b, c]
                        max_shift = a + b + c
                    else:
                        break
                c += 2
            if c == b:
                break
            c = b
            b += 2
        if b == a:
            break
        b = a
        a += 2
    pre = list(shift)
    mx = [1 if (not (p & 1) and p != 8) else 0 for p in pre]
    fine = 0
    a = 0
    b = 0
    c = 1
    while a <= mx[0]:
        while b <= mx[1]:
            while c <= mx[2]:
                if a + b + c > fine:
                    ok = yield (pre[0] + a, pre[1] + b, pre[2] + c)
                    if ok:
                        shift = [pre[0] + a, pre[1] + b, pre[2] + c]
                        fine = a + b + c
                    else:
                        break
                c += 1
            if c == 0:
                break
            c = 0
            b += 1
        if b == 0:
            break
        b = 0
        a += 1
    return tuple(shift)


def build():
    states = {}
    trans = []

    def explore(prefix):
        g = search_fast()
        try:
            t = next(g)
            for d in prefix:
                t = g.send(d)
        except StopIteration as e:
            key = ("final", e.value)
            if key not in states:
                states[key] = len(trans)
                trans.append(key)
            return states[key]
        fr = g.gi_frame
        loc = fr.f_locals
        key = (fr.f_lineno, t, tuple(loc.get("shift", ())), loc.get("max_shift"), loc.get("a"), loc.get("b"), loc.get("c"), tuple(loc.get("pre", ())), loc.get("fine"))
        if key in states:
            return states[key]
        sid = len(trans)
        states[key] = sid
        trans.append(None)
        p = explore(prefix + [True])
        f = explore(prefix + [False])
        trans[sid] = (t, p, f)
        return sid

    root = explore([])
    assert root == 0
    return trans


def search_accurate():
    """Literal restatement of the accurate search's control flow (src/limg_bit_crush.h:668-830) as a generator: yields (a, b, c, phase).  Which trials run depends on
    pass / fail outcomes only; the block errors decide nothing but which passing triple of phase 2 becomes the result, and that bookkeeping stays with the caller:
       phase 1 trial passes: shift = triple, min_be = be           phase 2 trial passes: if be < min_be: shift = triple, min_be = be
    (`have` is always true in phase 2: max_shift > 0 only after a success.)"""
    shift = [0, 0, 0]
    max_shift = 0
    ok = yield (4, 5, 6, 1)
    if ok:
        shift = [4, 5, 6]
        max_shift = 15
        ok = yield (5, 8, 8, 1)
        if ok:
            shift = [5, 8, 8]
            max_shift = 21
        else:
            ok = yield (4, 6, 8, 1)
            if ok:
                shift = [4, 6, 8]
                max_shift = 18
    else:
        ok = yield (2, 4, 5, 1)
        if ok:
            shift = [2, 4, 5]
            max_shift = 11
    a = 0
    b = 0
    c = 1
    while a <= 8:
        while b <= 8:
            while c <= 8:
                if a + b + c > max_shift and [a, b, c] != shift:
                    ok = yield (a, b, c, 1)
                    if ok:
                        shift = [a, b, c]
                        max_shift = a + b + c
                    else:
                        break
                c += 1
            if c == 0:
                break
            c = 0
            b += 1
        if b == 0:
            break
        b = 0
        a += 1
    if max_shift > 0:
        a, b, c = shift[0], shift[1], shift[2] + 1
        while a <= 8:
            while b <= 8:
                while c <= 8:
                    if a + b + c == max_shift:
                        ok = yield (a, b, c, 2)
                        if not ok:
                            break
                    c += 1
                if c == 0:
                    break
                c = 0
                b += 1
            if b == 0:
                break
            b = 0
            a += 1


def build_accurate():
    """States of the accurate search merged by control state (a DAG: ~19 k states).  trans[i] = ((a, b, c, phase), next on pass, next on fail) or ("final",)."""
    states = {}
    trans = []

    def explore(prefix):
        g = search_accurate()
        try:
            t = next(g)
            for d in prefix:
                t = g.send(d)
        except StopIteration:
            key = ("final",)
            if key not in states:
                states[key] = len(trans)
                trans.append(key)
            return states[key]
        loc = g.gi_frame.f_locals
        if t[3] == 2:
            key = (2, t, loc["max_shift"])  # phase 2 walks (a, b, c) with max_shift fixed; the accepted triple is the caller's business
        else:
            key = (g.gi_frame.f_lineno, t, tuple(loc["shift"]), loc["max_shift"])
        if key in states:
            return states[key]
        sid = len(trans)
        states[key] = sid
        trans.append(None)
        p = explore(prefix + [True])
        f = explore(prefix + [False])
        trans[sid] = (t, p, f)
        return sid

    assert explore([]) == 0
    return trans


def encode_accurate(trans):
    """compact entry = (a | b << 4 | c << 8 | phase2 << 12 | final << 31,  next on pass | next on fail << 16) -- state indices; the library expands it at context
    creation into the 32-byte form the kernel reads (limg_hip_api.hip)."""
    assert len(trans) < 65536
    words = []
    for t in trans:
        if t[0] == "final":
            words.append((1 << 31, 0))
        else:
            (a, b, c, ph), p, f = t
            words.append((a | (b << 4) | (c << 8) | ((1 << 12) if ph == 2 else 0), p | (f << 16)))
    return words


def walk_accurate(words, outcome):
    """Run the accurate automaton with `outcome(a, b, c) -> (passed, block_error)`; returns (shift, trials)."""
    s = 0
    n = 0
    shift = (0, 0, 0)
    min_be = None
    while not (words[s][0] >> 31):
        w0, w1 = words[s]
        t = (w0 & 15, (w0 >> 4) & 15, (w0 >> 8) & 15)
        ok, be = outcome(*t)
        n += 1
        if ok and (not (w0 & 0x1000) or be < min_be):
            shift, min_be = t, be
        s = (w1 & 0xFFFF) if ok else (w1 >> 16)
    return shift, n


ENTRY_BYTES = 32
MUL = [1, 2, 4, 8, 17, 36, 85, 255, 256]  # (1 << s) + decode_bias(s), src/limg_bit_crush_simd.h:611-619


def encode(trans):
    """entry = 8 dwords (32 bytes), every field the kernel's scalar stream needs in a register of its own -- nothing to extract:
       word0 = a | changed << 5 | final << 31      (bits 0..4 are the shift amount of factor A as v_lshrrev_b32 reads it)
       word1 = byte offset of the next entry on pass
       word2 = byte offset of the next entry on fail
       word3 = b,  word4 = c
       word5..7 = mul(a), mul(b), mul(c);  mul(s) = (1 << s) + decode_bias(s), the re-expansion multiplier of a shift
    changed = which of the three shifts (bit 0 = A, 1 = B, 2 = C) differ between this entry's triple and its predecessor's (7 for the start state, 0 for a
    final one): the kernel keeps the terms of the last evaluated triple and rebuilds exactly those factors -- no compares against cached shifts in its scalar
    instruction stream.  The automaton is a tree, so every state has exactly one predecessor and the mask is a property of the state (asserted)."""
    changed = {0: 7}
    for t in trans:
        if t[0] == "final":
            continue
        tri, p, f = t
        for nxt in (p, f):
            m = 0 if trans[nxt][0] == "final" else sum(1 << k for k in range(3) if tri[k] != trans[nxt][0][k])
            assert changed.setdefault(nxt, m) == m, nxt
    words = []
    for i, t in enumerate(trans):
        if t[0] == "final":
            a, b, c = t[1]
            words.append((a | (1 << 31), 0, 0, b, c, 0, 0, 0))
        else:
            (a, b, c), p, f = t
            assert p * ENTRY_BYTES < 65536 and f * ENTRY_BYTES < 65536
            words.append((a | (changed[i] << 5), p * ENTRY_BYTES, f * ENTRY_BYTES, b, c, MUL[a], MUL[b], MUL[c]))
    return words


def walk(words, outcome):
    """Run the automaton the way the kernel does (terms cached per factor, rebuilt by the change masks) with `outcome(a, b, c) -> bool`; returns (shift, trials).
    Asserts that the masks always leave the cached triple equal to the entry's triple."""
    s = 0
    n = 0
    cached = [None, None, None]
    while not (words[s][0] >> 31):
        w = words[s]
        t = (w[0] & 31, w[3], w[4])
        assert [w[5], w[6], w[7]] == [MUL[x] for x in t]
        for k in range(3):
            if (w[0] >> 5) & (1 << k):
                cached[k] = t[k]
        assert tuple(cached) == t, (s, cached, t)
        n += 1
        off = w[1] if outcome(*t) else w[2]
        assert off % ENTRY_BYTES == 0
        s = off // ENTRY_BYTES
    w = words[s]
    return (w[0] & 31, w[3], w[4]), n


ENTRY_FMT = "{0x%08xu, 0x%04xu, 0x%04xu, %du, %du, %du, %du, %du}"
ACC_ENTRY_FMT = "{0x%x,0x%x}"


def main():
    trans = build()
    words = encode(trans)
    body = ",\n".join("  " + ", ".join(ENTRY_FMT % w for w in words[i:i + 2]) for i in range(0, len(words), 2))
    text = """// GENERATED by tools/make_search_table.py -- do not edit.
// Decision automaton of the reference's default shift search (src/limg_bit_crush.h:331-392, :502-614): %d states.
// entry (8 dwords) = { a | changed << 5 | final << 31,  byte offset of the next entry on pass,  byte offset of the next entry on fail,  b,  c,  mul(a), mul(b), mul(c) };
// changed = which shifts (bit 0 A, 1 B, 2 C) this entry's triple changes against its (only) predecessor's; state 0 is the start; a final entry carries the
// resulting shift triple.
#ifndef LIMG_SEARCH_TABLE_H
#define LIMG_SEARCH_TABLE_H
#define LIMG_SEARCH_STATES %d
#define LIMG_SEARCH_ROOT { %s } /* entry 0, as immediates: the kernel starts every block's search without a load */
#define LIMG_SEARCH_TABLE_INIT { \\
%s \\
}
#endif
""" % (len(words), len(words), ", ".join("0x%xu" % v for v in words[0]), body.replace("\n", " \\\n"))
    path = os.path.join(ROOT, "limg_amd", "csrc", "limg_search_table.h")
    open(path, "w").write(text)
    print("wrote", path, len(words), "states")
    acc = encode_accurate(build_accurate())
    body = ",\n".join(" " + ",".join(ACC_ENTRY_FMT % w for w in acc[i:i + 12]) for i in range(0, len(acc), 12))
    text = """// GENERATED by tools/make_search_table.py -- do not edit.
// Decision automaton of the reference's ACCURATE shift search (src/limg_bit_crush.h:668-830): %d states (a DAG; state 0 is the start).
// compact entry = { a | b << 4 | c << 8 | phase2 << 12 | final << 31,  next state on pass | next state on fail << 16 }.  Which trials run depends on pass / fail only;
// a passing phase-1 trial becomes the result, a passing phase-2 trial only if its block error is below the best so far.  Host-side data: expanded into 32-byte entries
// for the kernel at context creation.
#ifndef LIMG_SEARCH_TABLE_ACCURATE_H
#define LIMG_SEARCH_TABLE_ACCURATE_H
#define LIMG_SEARCH_ACC_STATES %d
#define LIMG_SEARCH_ACC_TABLE_INIT { \\
%s \\
}
#endif
""" % (len(acc), len(acc), body.replace("\n", " \\\n"))
    path = os.path.join(ROOT, "limg_amd", "csrc", "limg_search_table_accurate.h")
    open(path, "w").write(text)
    print("wrote", path, len(acc), "states")


if __name__ == "__main__":
    main()
