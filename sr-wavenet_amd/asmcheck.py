"""Build-time check of hand-counted register loads in gfx950 assembly (used by build.py for the kernels of NO_SPILL).

A kernel that fetches register operands with inline-asm ``global_load`` and retires them with a hand-counted
``s_waitcnt vmcnt(N)`` hides those loads from the compiler: it believes the destination written when the asm statement
ends, so it may copy, re-coalesce or reuse the register while the data is still in flight (seen on a prototype as
HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION: a destination reused for an address).  A spill count of zero does not exclude
that.  This walks the kernel's assembly instead:

* every ``global_load_*`` / ``buffer_load_*`` / ``scratch_load_*`` with a VGPR destination opens an in-flight entry;
* every vector-memory operation issued after it (loads, stores, atomics, LDS-DMA: they all count in ``vmcnt``, in issue
  order) ages it by one; an ``s_waitcnt vmcnt(N)`` retires the entries that have at least N younger operations;
* until then no instruction may read or write any register of the destination.

The walk is a forward dataflow over the kernel's basic blocks (labels / branches), merging at joins with the union of
the entries and the MINIMUM of their ages (an operation on one path only cannot be relied on to have been issued), to a
fixpoint -- loops included.  Returns the list of violations (empty = clean).
"""
from __future__ import annotations

import re
from typing import Dict, List, Tuple

_REG = re.compile(r"\b([va])(?:(\d+)|\[(\d+):(\d+)\])")
_VM_PREFIX = ("global_load", "global_store", "global_atomic", "buffer_load", "buffer_store", "buffer_atomic",
              "buffer_wbl2", "buffer_inv", "scratch_load", "scratch_store", "flat_load", "flat_store", "flat_atomic",
              "image_", "tbuffer_")
_LOAD_PREFIX = ("global_load", "buffer_load", "scratch_load", "flat_load")
AGE_CAP = 64      # vmcnt is a 6-bit counter


def _regs(text: str) -> set:
    out = set()
    for m in _REG.finditer(text):
        f = m.group(1)
        if m.group(2) is not None:
            out.add((f, int(m.group(2))))
        else:
            for r in range(int(m.group(3)), int(m.group(4)) + 1):
                out.add((f, r))
    return out


def _vmcnt_of(ins: str):
    """vmcnt an ``s_waitcnt`` waits for, or None if it leaves vmcnt alone."""
    m = re.search(r"vmcnt\((\d+)\)", ins)
    if m:
        return int(m.group(1))
    m = re.match(r"s_waitcnt\s+(0x[0-9a-fA-F]+|\d+)\s*$", ins)
    if m:      # raw simm16 (gfx9): vmcnt = bits 3:0 | bits 15:14 << 4
        v = int(m.group(1), 0)
        return (v & 0xF) | (((v >> 14) & 3) << 4)
    return None


def function_body(asm: str, kernel: str) -> List[str]:
    """Instruction / label lines of the first function whose symbol contains `kernel`."""
    lines = asm.splitlines()
    start = None
    for i, ln in enumerate(lines):
        m = re.match(r"^([A-Za-z_$][\w$.]*):", ln)
        if m and kernel in m.group(1) and not m.group(1).startswith(".L"):
            start = i + 1
            break
    if start is None:
        raise ValueError("kernel %r not found in the assembly" % kernel)
    body = []
    for ln in lines[start:]:
        s = ln.split(";", 1)[0].strip()
        if s.startswith(".Lfunc_end"):
            break
        if not s or (s.startswith(".") and not s.endswith(":")):
            continue      # directives
        body.append(s)
    return body


def check_inflight_loads(asm: str, kernel: str) -> List[str]:
    body = function_body(asm, kernel)
    # ---- basic blocks
    leaders = {0}
    label_at: Dict[str, int] = {}
    for i, s in enumerate(body):
        if s.endswith(":"):
            label_at[s[:-1]] = i
            leaders.add(i)
        elif s.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc", "s_swappc")):
            leaders.add(i + 1)
    starts = sorted(x for x in leaders if x < len(body))
    block_of = {}
    blocks: List[Tuple[int, int]] = []
    for n, st in enumerate(starts):
        en = starts[n + 1] if n + 1 < len(starts) else len(body)
        blocks.append((st, en))
        block_of[st] = n
    succ: List[List[int]] = []
    for n, (st, en) in enumerate(blocks):
        last = body[en - 1]
        out = []
        if last.startswith("s_endpgm"):
            pass
        elif last.startswith("s_branch"):
            out.append(block_of[label_at[last.split()[1]]])
        else:
            if last.startswith("s_cbranch"):
                out.append(block_of[label_at[last.split()[1]]])
            if en < len(body):
                out.append(block_of[en])
        succ.append(out)

    Entry = Dict[frozenset, int]      # destination registers -> minimum number of younger vector-memory operations

    def transfer(state: Entry, st: int, en: int, report: List[str]) -> Entry:
        cur = dict(state)
        for i in range(st, en):
            s = body[i]
            if s.endswith(":"):
                continue
            op = s.split()[0]
            n = _vmcnt_of(s) if op == "s_waitcnt" else None
            if n is not None:
                cur = {d: a for d, a in cur.items() if a < n}
                continue
            touched = _regs(s)
            for d in cur:
                if touched & d:
                    report.append("line %d: `%s` touches %s of a load still in flight (>= %d younger operations, no covering wait)"
                                  % (i, s, sorted(touched & d), cur[d]))
            if op.startswith(_VM_PREFIX):
                cur = {d: min(a + 1, AGE_CAP) for d, a in cur.items()}
                if op.startswith(_LOAD_PREFIX) and "lds" not in op.split("_"):
                    ops = s[len(op):].split(",")
                    dest = frozenset(_regs(ops[0])) if ops else frozenset()
                    if dest and " lds" not in s:
                        # (a reload into registers already in flight is itself reported above as a touch)
                        cur = {d: a for d, a in cur.items() if not (d & dest)}
                        cur[dest] = 0
        return cur

    def merge(a: Entry, b: Entry) -> Entry:
        out = dict(a)
        for d, age in b.items():
            out[d] = min(out[d], age) if d in out else age
        return out

    ins: List[Entry] = [None] * len(blocks)
    ins[0] = {}
    work = [0]
    while work:
        n = work.pop()
        out = transfer(ins[n], blocks[n][0], blocks[n][1], [])
        for m in succ[n]:
            new = out if ins[m] is None else merge(ins[m], out)
            if ins[m] is None or new != ins[m]:
                ins[m] = new
                work.append(m)
    report: List[str] = []
    for n, (st, en) in enumerate(blocks):
        if ins[n] is not None:
            transfer(ins[n], st, en, report)
    # an instruction inside a loop is visited once per block: no duplicates to remove, but keep the order stable
    return sorted(set(report), key=lambda r: int(r.split()[1].rstrip(":")))


if __name__ == "__main__":
    import sys
    bad = check_inflight_loads(open(sys.argv[1]).read(), sys.argv[2])
    print("\n".join(bad) if bad else "clean")
    sys.exit(1 if bad else 0)
