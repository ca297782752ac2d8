"""Static scan of a gfx950 disassembly for the hazard behind round 4's 'two-rank' discrepancy: a VGPR that is the
destination of an LDS read still in flight — ds_read* issued from inline asm, which the compiler's own s_waitcnt insertion does not
see — being READ or WRITTEN by another instruction before an s_waitcnt lgkmcnt covers it.  The hardware does not interlock VGPR accesses
against outstanding LDS returns: such an instruction sees the register's old contents whenever the LDS pipeline is slow (for
instance under another process's LDS-bound waves on the same CU).

    llvm-objdump -d --mcpu=gfx950 <code object> | python tools/diag/lds_hazard_scan.py [kernel-name-substring]

Model: lgkmcnt counts LDS (and scalar-memory) operations; LDS operations return in order, so `s_waitcnt lgkmcnt(N)` retires all but the
N youngest.  The scan is a forward may-analysis over the function's control-flow graph (branch targets from the s_branch / s_cbranch
immediates); at a join the predecessors' queues are merged aligned at their youngest ends.  `scan(lines)` is importable (the test
suite runs it over the built library)."""
import re
import subprocess
import sys

reg_re = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")
lg_re = re.compile(r"lgkmcnt\((\d+)\)")
vm_re = re.compile(r"vmcnt\((\d+)\)")
ins_re = re.compile(r"^\s+(\S+)(?:\s+(.*?))?\s*//\s*([0-9A-Fa-f]+):")


def regs(tok):
    out = set()
    for m in reg_re.finditer(tok):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            for k in range(int(m.group(4)), int(m.group(5)) + 1):
                out.add((m.group(3), k))
    return out


def _merge(a, b):
    """two queues of (dst-set, line, text), aligned at the youngest end; the longer one's older entries are kept"""
    if a is None:
        return b
    if len(a) < len(b):
        a, b = b, a
    out = list(a)
    off = len(a) - len(b)
    changed = False
    for i, (d, ln, t) in enumerate(b):
        d0, l0, t0 = out[off + i]
        if not d <= d0:
            out[off + i] = (d0 | d, l0, t0)
            changed = True
    return tuple(out), changed or len(a) != len(b)


def analyse_function(ins, vm=False):
    """ins: list of (addr, op, args, line_no, text) -> list of (line_no, text, (pending line, pending text)).
    vm: also model vmcnt (vector-memory loads into VGPRs).  Off by default: the analysis is path-insensitive, and the compiler lowers
    `if (c) s_waitcnt vmcnt(a) else s_waitcnt vmcnt(b)` to two correlated branches whose infeasible combination (neither wait) shows up
    as false positives in the software-pipelined loops of gemm_wd.hip; moves_of_loaded_registers() covers that file instead."""
    index = {a: i for i, (a, *_rest) in enumerate(ins)}
    n = len(ins)
    succ = [[] for _ in range(n)]
    for i, (addr, op, args, ln, text) in enumerate(ins):
        nxt = i + 1 if i + 1 < n else None
        if op in ("s_endpgm", "s_setpc_b64", "s_swappc_b64"):
            continue
        if op.startswith("s_cbranch") or op == "s_branch":
            try:
                imm = int(args.split()[0])
            except (ValueError, IndexError):
                imm = None
            if imm is not None:
                if imm >= 32768:
                    imm -= 65536
                tgt = addr + 4 + 4 * imm
                if tgt in index:
                    succ[i].append(index[tgt])
            if op != "s_branch" and nxt is not None:
                succ[i].append(nxt)
            continue
        if nxt is not None:
            succ[i].append(nxt)
    # state = (lgkm queue, vm queue): LDS / scalar-memory operations counted by lgkmcnt, vector-memory operations counted by vmcnt
    # (gfx9: loads and stores share the counter and retire in issue order — the compiler's own model)
    state_in = [None] * n
    state_in[0] = ((), ())
    work = [0]
    hazards = {}
    steps = 0
    while work and steps < 60 * n + 1000:
        steps += 1
        i = work.pop()
        q, v = list(state_in[i][0]), list(state_in[i][1])
        addr, op, args, ln, text = ins[i]

        def pending(used):
            for dst, l0, t0 in q:
                if dst & used:
                    return l0, t0
            for dst, l0, t0 in v:
                if dst & used:
                    return l0, t0
            return None

        if op.startswith("s_waitcnt"):
            m = lg_re.search(args)
            if m:
                k = int(m.group(1))
                q = q[len(q) - k:] if k > 0 else []
            m = vm_re.search(args)
            if m:
                k = int(m.group(1))
                v = v[len(v) - k:] if k > 0 else []
            if "cnt" not in args:
                q, v = [], []
        else:
            is_lds = op.startswith("ds_")
            is_smem = op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_memtime") or op.startswith("s_memrealtime")
            is_vmem = vm and op.startswith(("global_", "buffer_", "flat_", "scratch_", "tbuffer_", "image_"))
            if is_lds or is_smem:
                toks = args.split(",")
                reads = is_lds and (op.startswith("ds_read") or op.startswith("ds_load") or "permute" in op or op.startswith("ds_swizzle") or "rtn" in op)
                dst = regs(toks[0]) if reads else set()
                used = regs(",".join(toks[1:] if reads else toks))
                hit = pending(used)              # (a re-load into a register whose earlier load is pending is safe: same counter, in-order return)
                if hit:
                    hazards[ln] = (ln, text, hit)
                q.append((frozenset(dst), ln, text))
                if len(q) > 64:
                    q = q[-64:]
            elif is_vmem:
                toks = args.split(",")
                to_lds = "_lds_" in op or " lds" in args
                loads = ("load" in op and not to_lds) or ("atomic" in op and ("sc0" in args or "glc" in args))
                dst = regs(toks[0]) if loads else set()
                used = regs(",".join(toks[1:] if loads else toks))
                hit = pending(used)
                if hit:
                    hazards[ln] = (ln, text, hit)
                v.append((frozenset(dst), ln, text))
                if len(v) > 64:
                    v = v[-64:]
            elif q or v:
                used = regs(args)
                hit = pending(used) if used else None
                if hit:
                    hazards[ln] = (ln, text, hit)
        out = (tuple(q), tuple(v))
        for j in succ[i]:
            if state_in[j] is None:
                state_in[j] = out
                work.append(j)
            else:
                mq, cq = _merge(state_in[j][0], out[0])
                mv, cv = _merge(state_in[j][1], out[1])
                if cq or cv:
                    state_in[j] = (mq, mv)
                    work.append(j)
    return [hazards[k] for k in sorted(hazards)]


def moves_of_loaded_registers(lines, name_filter):
    """for the functions whose (mangled) name contains name_filter: register-to-register moves (v_mov*, v_accvgpr_*) whose source is,
    anywhere in the function, the destination of a global_load into VGPRs.  The weight-direct GEMM issues those loads from inline asm
    and keeps D K-tiles of them in flight across its loop's back edge; its counted vmcnt waits are only valid while the compiler never
    copies such a register (a copy would read it before our wait).  -> {name: [text, ...]}"""
    out = {}
    name, ins = None, []

    def flush():
        if name is None or name_filter not in name:
            return
        loaded = set()
        for op, args in ins:
            if op.startswith("global_load_dword") and "lds" not in op:
                loaded |= regs(args.split(",")[0])
        bad = [f"{op} {args}" for op, args in ins
               if op.startswith(("v_mov_b", "v_accvgpr_write", "v_accvgpr_mov", "v_pk_mov")) and regs(",".join(args.split(",")[1:])) & loaded]
        out[name] = bad

    for line in lines:
        m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
        if m:
            flush()
            name, ins = m.group(1), []
            continue
        m = ins_re.match(line)
        if name is not None and m:
            ins.append((m.group(1), m.group(2) or ""))
    flush()
    return out


def scan(lines, vm=False):
    """-> {mangled function name: [(line no, text, (pending since line, text)), ...]} for functions with at least one hazard"""
    report = {}
    name, ins = None, []

    def flush():
        if name is not None and ins:
            hz = analyse_function(ins, vm)
            if hz:
                report[name] = hz

    for ln, line in enumerate(lines, 1):
        m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
        if m:
            flush()
            name, ins = m.group(1), []
            continue
        if name is None:
            continue
        m = ins_re.match(line)
        if not m:
            continue
        ins.append((int(m.group(3), 16), m.group(1), m.group(2) or "", ln, line.strip()))
    flush()
    return report


def demangle(s):
    try:
        return subprocess.run(["c++filt", s], capture_output=True, text=True).stdout.strip()
    except Exception:  # noqa: BLE001
        return s


if __name__ == "__main__":
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    rep = scan(sys.stdin)
    shown = 0
    for k, v in rep.items():
        dn = demangle(k)
        if flt and flt not in dn:
            continue
        shown += 1
        print(f"== {dn[:200]}: {len(v)} accesses to a VGPR with an LDS read in flight")
        for ln, text, src in v[:8]:
            print(f"   line {ln}: {text[:110]}   <- pending since {src[0]}: {src[1][:80]}")
    print(f"{sum(len(v) for v in rep.values())} in {len(rep)} functions")
