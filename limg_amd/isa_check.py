"""The check that keeps k_stream_decode's hand-counted wait honest, run by limg_amd/build.py ON THE COMPILE LINE THAT SHIPS (and by tests/test_decode_isa.py).

k_stream_decode requests its payload runs with inline-assembly loads and waits for them with a hand-counted `s_waitcnt vmcnt(2)` (limg_hip_stream.hip: the compiler's
own wait for a load consumed across the loop's back edge is vmcnt(0), which also drains the group's stores).  The compiler therefore believes the loaded registers are
valid from the asm statement on.  Given the kernel's assembly, check_decode_isa verifies the things that belief must not break: no instruction reads a register pair with a
load in flight before the next hand-written wait, every path from those loads to the wait is free of scratch traffic (a spill reload would count in vmcnt and void the
count), and on every path at least N vector-memory operations lie between the last hand-issued load and a hand-written `s_waitcnt vmcnt(N)`.  Raises AssertionError."""
import re


def check_decode_isa(asm_text):
    text = asm_text.splitlines()
    start = next(i for i, l in enumerate(text) if l.startswith("_ZN8limg_hip12_GLOBAL__N_115k_stream_decodeE"))
    end = next(i for i in range(start, len(text)) if text[i].startswith(".Lfunc_end"))
    body = text[start:end]
    assert not any("scratch_" in l for l in body), "k_stream_decode spills: a scratch reload counts in vmcnt and voids the hand-counted wait"
    # basic blocks (labels, branches) and a forward data-flow of "register pairs with a hand-issued load in flight" over them: the loads of a run's second and third
    # piece sit in out-of-line blocks, so a linear scan would not do
    blocks, order, cur, in_asm = {}, [], "entry", False
    blocks[cur] = []
    order.append(cur)
    for l in body[1:]:
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            order.append(cur)
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        blocks[cur].append((t.split(";")[0].strip(), in_asm))
    succ = {}
    for i, b in enumerate(order):
        out, fall = [], True
        for t, _ in blocks[b]:
            m = re.match(r"s_(cbranch_\w+|branch) (\.LBB\d+_\d+)", t)
            if m:
                out.append(m.group(2))
                if m.group(1) == "branch":
                    fall = False
            if t.startswith("s_endpgm"):
                fall = False
        if fall and i + 1 < len(order):
            out.append(order[i + 1])
        succ[b] = out
    loads = waits = 0
    state = {b: None for b in order}
    state["entry"] = frozenset()
    work = ["entry"]
    while work:
        b = work.pop()
        pending = set(state[b])
        for t, asm in blocks[b]:
            if asm and t.startswith("global_load_dwordx2"):
                m = re.match(r"global_load_dwordx2 v\[(\d+):(\d+)\]", t)
                assert m, t
                pending |= {int(m.group(1)), int(m.group(2))}
                loads += 1
                continue
            if asm and t.startswith("s_waitcnt vmcnt("):
                pending.clear()
                waits += 1
                continue
            if t.startswith("s_cbranch") or t.startswith("s_branch"):
                continue
            if pending:
                regs = set(int(x) for x in re.findall(r"\bv(\d+)\b", t))
                for a0, b0 in re.findall(r"v\[(\d+):(\d+)\]", t):
                    regs |= set(range(int(a0), int(b0) + 1))
                assert not (regs & pending), "register with a payload load in flight is touched before the counted wait: %s (in flight: v%s, block %s)" % (t, sorted(pending), b)
        for n in succ[b]:
            new = frozenset(pending) if state[n] is None else state[n] | frozenset(pending)
            if new != state[n]:
                state[n] = new
                work.append(n)
    assert loads >= 6 and waits >= 2, (loads, waits)  # three loads at a unit's top, three in the group loop; the full wait and the counted one

    # The count itself: on EVERY path from a hand-issued load to a hand-written `s_waitcnt vmcnt(N)`, at least N vector-memory operations must have been issued after the
    # last such load (vmcnt retires in order: then "at most N outstanding" implies the loads are back).  Minimum over paths of the operations since the last load.
    INF = 1 << 30
    cnt = {b: None for b in order}
    cnt["entry"] = INF
    work = ["entry"]
    checked = 0
    while work:
        b = work.pop()
        c = cnt[b]
        for t, asm in blocks[b]:
            if asm and t.startswith("global_load_dwordx2"):
                c = 0
            elif asm and t.startswith("s_waitcnt vmcnt("):
                n = int(re.match(r"s_waitcnt vmcnt\((\d+)\)", t).group(1))
                assert c >= n, "a path reaches `%s` with only %d vector-memory operations behind the payload loads (block %s)" % (t, c, b)
                checked += 1
                c = INF
            elif re.match(r"(global|buffer|flat|scratch)_(load|store|atomic)", t) and c != INF:
                c += 1
        for nb in succ[b]:
            new = c if cnt[nb] is None else min(cnt[nb], c)
            if new != cnt[nb]:
                cnt[nb] = new
                work.append(nb)
    assert checked >= 2
    return {"loads": loads, "waits": waits, "counted_waits": checked}
