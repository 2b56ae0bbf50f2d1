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


def encode(trans):
    """entry = (word0, word1):
       word0 = a | b << 4 | c << 8 | changed_on_pass << 12 | changed_on_fail << 15 | final << 31
       word1 = byte offset of the next entry on pass | byte offset on fail << 16          (entries are 16 bytes)
       word2 = mul(a) | mul(b) << 9 | mul(c) << 18, mul(s) = (1 << s) + decode_bias(s): the re-expansion multiplier of each shift, so that the kernel's scalar
               stream extracts it with one bit-field instruction instead of computing it (5 instructions)
    changed_on_* = which of the three shifts (bit 0 = A, 1 = B, 2 = C) differ between this entry's triple and the successor's: the kernel keeps the terms of the
    last evaluated triple and rebuilds exactly those factors -- no compares against cached shifts in its scalar instruction stream.  0 for a final successor."""
    MUL = [1, 2, 4, 8, 17, 36, 85, 255, 256]  # (1 << s) + decode_bias(s), src/limg_bit_crush_simd.h:611-619

    def muls(t):
        return MUL[t[0]] | (MUL[t[1]] << 9) | (MUL[t[2]] << 18)

    def changed(t, nxt):
        if nxt[0] == "final":
            return 0
        u = nxt[0]
        return sum(1 << k for k in range(3) if t[k] != u[k])
    words = []
    for t in trans:
        if t[0] == "final":
            a, b, c = t[1]
            words.append((a | (b << 4) | (c << 8) | (1 << 31), 0, 0))
        else:
            (a, b, c), p, f = t
            assert p * 16 < 65536 and f * 16 < 65536
            words.append((a | (b << 4) | (c << 8) | (changed((a, b, c), trans[p]) << 12) | (changed((a, b, c), trans[f]) << 15), (p * 16) | ((f * 16) << 16), muls((a, b, c))))
    return words


def walk(words, outcome):
    """Run the automaton the way the kernel does (terms cached per factor, rebuilt by the change masks) with `outcome(a, b, c) -> bool`; returns (shift, trials).
    Asserts that the masks always leave the cached triple equal to the entry's triple."""
    s = 0
    n = 0
    cached = [None, None, None]
    chg = 7
    while not (words[s][0] >> 31):
        w0, w1, w2 = words[s]
        t = (w0 & 15, (w0 >> 4) & 15, (w0 >> 8) & 15)
        assert [(w2 >> (9 * k)) & 511 for k in range(3)] == [[1, 2, 4, 8, 17, 36, 85, 255, 256][x] for x in t]
        for k in range(3):
            if chg & (1 << k):
                cached[k] = t[k]
        assert tuple(cached) == t, (s, cached, t)
        n += 1
        ok = outcome(*t)
        chg = (w0 >> (12 if ok else 15)) & 7
        off = (w1 & 0xFFFF) if ok else (w1 >> 16)
        assert off % 16 == 0
        s = off // 16
    w0 = words[s][0]
    return (w0 & 15, (w0 >> 4) & 15, (w0 >> 8) & 15), n


def main():
    trans = build()
    words = encode(trans)
    body = ",\n".join("  " + ", ".join("{0x%08xu, 0x%08xu, 0x%08xu, 0u}" % w for w in words[i:i + 4]) for i in range(0, len(words), 4))
    text = """// GENERATED by tools/make_search_table.py -- do not edit.
// Decision automaton of the reference's default shift search (src/limg_bit_crush.h:331-392, :502-614): %d states.
// entry (16 bytes) = { a | b << 4 | c << 8 | changed_on_pass << 12 | changed_on_fail << 15 | final << 31,  byte offset of the next entry on pass | on fail << 16,
//                     mul(a) | mul(b) << 9 | mul(c) << 18,  0 };
// changed_on_* = which shifts (bit 0 A, 1 B, 2 C) the successor's triple changes; state 0 is the start; a final entry carries the resulting shift triple.
#ifndef LIMG_SEARCH_TABLE_H
#define LIMG_SEARCH_TABLE_H
#define LIMG_SEARCH_STATES %d
#define LIMG_SEARCH_ROOT_X 0x%08xu /* entry 0, as immediates: the kernel starts every block's search without a load */
#define LIMG_SEARCH_ROOT_Y 0x%08xu
#define LIMG_SEARCH_ROOT_Z 0x%08xu
#define LIMG_SEARCH_TABLE_INIT { \\
%s \\
}
#endif
""" % (len(words), len(words), words[0][0], words[0][1], words[0][2], body.replace("\n", " \\\n"))
    path = os.path.join(ROOT, "limg_amd", "csrc", "limg_search_table.h")
    open(path, "w").write(text)
    print("wrote", path, len(words), "states")


if __name__ == "__main__":
    main()
