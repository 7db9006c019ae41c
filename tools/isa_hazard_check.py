"""Static check of the generated gfx950 code for the one hazard hand-issued LDS reads bring: `asm volatile("ds_read_b128 %0, ...")` returns at
once as far as the compiler knows, the data lands later, and only our own `s_waitcnt lgkmcnt(0)` (tile_prims.h: frag_wait) orders a use behind it.
Nothing stops the register allocator from COPYING such a register (a phi on a control-flow edge, a split live range) between the read and the
wait -- it then copies what the register held before.  This script compiles a .hip file to assembly, builds each kernel's control-flow graph
and reports every instruction that touches a register with a hand-issued LDS read still in flight.

    python tools/isa_hazard_check.py aldi_amd/csrc/igemm.hip [more.hip ...]        exit status 1 if anything is found
"""
import os
import re
import subprocess
import sys
import tempfile

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S", "--cuda-device-only"]
REG = re.compile(r"\b([va])(?:(\d+)|\[(\d+):(\d+)\])")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        kind = m.group(1)
        if m.group(2) is not None:
            out.add((kind, int(m.group(2))))
        else:
            out.update((kind, i) for i in range(int(m.group(3)), int(m.group(4)) + 1))
    return out


def parse_kernels(asm_text):
    """-> {kernel name: [(label or None, instruction text, in_asm_block)]}"""
    kernels, cur, name, in_asm = {}, None, None, False
    for line in asm_text.splitlines():
        s = line.strip()
        m = re.match(r"^([A-Za-z_][\w$.]*):", line)
        if m and not line.startswith(".L") and cur is None and (m.group(1).startswith("_Z") or "kernel" in m.group(1)):
            name, cur = m.group(1), []
            continue
        if cur is None:
            continue
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        lm = re.match(r"^(\.LBB[\w]*):", line)
        if lm:
            cur.append((lm.group(1), None, False))
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        ins = s.split(";")[0].strip()
        if ins:
            cur.append((None, ins, in_asm))
        if ins.startswith("s_endpgm"):
            kernels[name] = cur
            cur, name = None, None
    return kernels


def check_kernel(items):
    # basic blocks
    blocks, order, cur = {}, [], "entry"
    blocks[cur] = []
    order.append(cur)
    for label, ins, in_asm in items:
        if label is not None:
            blocks.setdefault(label, [])
            if label not in order:
                order.append(label)
            cur = label
            continue
        blocks[cur].append((ins, in_asm))
    succ = {b: [] for b in order}
    for i, b in enumerate(order):
        fall = True
        for ins, _ in blocks[b]:
            m = re.match(r"^(s_branch|s_cbranch_\w+)\s+(\.LBB\w+)", ins)
            if m:
                if m.group(2) in succ:
                    succ[b].append(m.group(2))
                if m.group(1) == "s_branch":
                    fall = False
            if ins.startswith("s_endpgm"):
                fall = False
        if fall and i + 1 < len(order):
            succ[b].append(order[i + 1])

    CAP = 96

    def transfer(b, state, report=None):
        # LDS operations return IN ORDER: `s_waitcnt lgkmcnt(N)` leaves the N youngest of them in flight (the interleaved loops of r06 wait with
        # counts).  The state is the QUEUE of LDS operations in flight, oldest first, each with the registers a hand-issued read will write
        # (compiler-issued LDS / scalar-memory operations take a slot with no tracked registers).
        q = list(state)
        for ins, in_asm in blocks[b]:
            op = ins.split()[0]
            m = re.search(r"lgkmcnt\((\d+)\)", ins) if op == "s_waitcnt" else None
            if m:
                n = int(m.group(1))
                q = q[len(q) - n:] if (n and n < len(q)) else ([] if n == 0 else q)
                continue
            touched = regs_of(ins[len(op):])
            inflight = set().union(*q) if q else set()
            if op.startswith("ds_read") and in_asm:
                dst = regs_of(ins[len(op):].split(",")[0])
                bad = (touched - dst) & inflight
                if bad and report is not None:
                    report.append((b, ins, sorted(bad)))
                q.append(frozenset(dst))
            else:
                bad = touched & inflight
                if bad and report is not None:
                    report.append((b, ins, sorted(bad)))
                if op.startswith("ds_") or op.startswith("s_load") or op.startswith("s_buffer_load"):
                    q.append(frozenset())
            if len(q) > CAP:
                q = [frozenset().union(*q[: len(q) - CAP + 1])] + q[len(q) - CAP + 1:]
        return tuple(q)

    def merge(x, y):
        # two queues reaching one block: aligned at the YOUNGEST end (what a counted wait counts from), entry-wise unions; the surplus of the
        # longer one joins the oldest entry
        if x == y:
            return x
        if len(x) < len(y):
            x, y = y, x
        k = len(y)
        if k == 0:
            return (frozenset().union(*x),) if x else ()
        head = x[: len(x) - k]
        out = [x[len(x) - k + i] | y[i] for i in range(k)]
        if head:
            out[0] = out[0] | frozenset().union(*head)
        return tuple(out)
    IN = {b: None for b in order}
    IN[order[0]] = ()
    changed, rounds = True, 0
    while changed and rounds < 200:
        changed = False
        rounds += 1
        for b in order:
            if IN[b] is None:
                continue
            out = transfer(b, IN[b])
            for t in succ[b]:
                new = out if IN[t] is None else merge(IN[t], out)
                if new != IN[t]:
                    IN[t] = new
                    changed = True
    for b in order:
        if IN[b] is None:
            IN[b] = ()
    report = []
    for b in order:
        transfer(b, IN[b], report)
    return report


_ASM = {}


def assembly_of(path):
    """the gfx950 assembly of one .hip file (compiled once per process)"""
    if path not in _ASM:
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "k.s")
            subprocess.check_call([HIPCC] + FLAGS + [path, "-o", out], stderr=subprocess.DEVNULL)
            _ASM[path] = open(out).read()
    return _ASM[path]


def scratch_sizes(path):
    """-> {kernel symbol: bytes of scratch per lane} (the `.amdhsa_private_segment_fixed_size` of each kernel descriptor).  The hot kernels with
    hand-placed counted waits sit within a few registers of their budget: a spill puts scratch loads / stores -- vector-memory operations the
    counted `vmcnt` waits do not know about -- into their loops (the 256 x 256 halo tile ran 30 % slower when its ablation flags became
    compile-time constants and the allocator, with more freedom, spilled 24 registers)."""
    out, name = {}, None
    for line in assembly_of(path).splitlines():
        m = re.match(r"^\s*\.amdhsa_kernel\s+(\S+)", line)
        if m:
            name = m.group(1)
        m = re.match(r"^\s*\.amdhsa_private_segment_fixed_size\s+(\d+)", line)
        if m and name:
            out[name] = int(m.group(1))
    return out


def check_file(path, keep=None):
    text = assembly_of(path)
    if keep:
        open(keep, "w").write(text)
    found = {}
    for name, items in parse_kernels(text).items():
        if not any(in_asm and ins.startswith("ds_read") for _, ins, in_asm in items if ins):
            continue
        rep = check_kernel(items)
        found[name] = rep
    return found


def main():
    rc = 0
    for path in sys.argv[1:]:
        res = check_file(path)
        n_bad = sum(1 for r in res.values() if r)
        print(f"{path}: {len(res)} kernels with hand-issued LDS reads, {n_bad} with a register touched before its wait")
        for name, rep in res.items():
            for b, ins, regs in rep[:12]:
                print(f"  {name[:90]}  {b}: {ins}   <- in flight: {regs[:8]}")
            if rep:
                rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main())
