"""Static scan of a gfx950 disassembly for the pattern behind round 5's farthest-point discrepancy: the result of a packed
fp32 instruction (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: 64-bit destination, written through the destination-select path) consumed by
the NEXT vector instruction of the wave with only scalar instructions in between.  The compiler's hazard recognizer counts any instruction
as the one wait state the forwarding hazard needs and fills the slot with an s_add / s_addc it had to place anyway; on the MI355X that is
not always enough when waves of another kernel share the SIMD — the consumer's upper lanes then see the register's OLD contents (observed:
fps_kernel's running minimum of the lane's first point, lanes 52-61, once in 10^2 .. 10^4 launches next to a GEMM).  An `s_nop` in the slot
(what the compiler emits when it has nothing else to place) was never seen to fail.

    llvm-objdump -d --mcpu=gfx950 <code object> | python tools/diag/pk_hazard_scan.py

scan(lines) -> {kernel: [(line_no, producer text, consumer text, fillers)]}"""
import re
import sys

ins_re = re.compile(r"^\s+(\S+)(?:\s+(.*?))?\s*//\s*([0-9A-Fa-f]+):")
fn_re = re.compile(r"^[0-9a-fA-F]+ <(.+)>:")
reg_re = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
PRODUCERS = ("v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32")


def vregs(tok):
    out = set()
    for m in reg_re.finditer(tok):
        if m.group(1):
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def scan(lines):
    rep = {}
    fn = None
    ins = []

    def flush():
        if fn is None:
            return
        for i, (op, args, ln, text) in enumerate(ins):
            if op not in PRODUCERS:
                continue
            dst = vregs(args.split(",")[0])
            fill = []
            for op2, args2, ln2, text2 in ins[i + 1:i + 12]:
                if op2.startswith("v_") or op2.startswith("ds_") or op2.startswith("global_") or op2.startswith("buffer_") or op2.startswith("flat_"):
                    srcs = vregs(",".join(args2.split(",")[1:])) if op2.startswith("v_") else vregs(args2)
                    if op2.startswith("v_") and srcs & dst and fill and not any(f.startswith("s_nop") for f in fill):
                        rep.setdefault(fn, []).append((ln, text.strip()[:70], text2.strip()[:70], fill))
                    break
                if op2.startswith("s_cbranch") or op2.startswith("s_branch") or op2 in ("s_endpgm", "s_barrier", "s_waitcnt"):
                    break
                fill.append(op2)

    for ln, line in enumerate(lines, 1):
        m = fn_re.match(line)
        if m:
            flush()
            fn, ins = m.group(1), []
            continue
        m = ins_re.match(line)
        if m:
            ins.append((m.group(1), m.group(2) or "", ln, line))
    flush()
    return rep


if __name__ == "__main__":
    rep = scan(sys.stdin.readlines())
    n = 0
    for k, v in rep.items():
        print(f"{k[:110]}: {len(v)}")
        for h in v[:4]:
            print("   ", h[1], "->", h[2], h[3])
        n += len(v)
    print(f"{n} packed-fp32 results consumed behind scalar fillers only, in {len(rep)} kernels")
