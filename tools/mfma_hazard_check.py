"""Static check of a hipcc -S dump for the hazards an INLINE-ASM MFMA hides from the compiler (its hazard recogniser and its
waitcnt pass do not look inside asm blocks).  Three classes, each listed with kernel and instruction:

  A  a VALU instruction writes a source VGPR of an asm MFMA within the two preceding instructions (no wait states are inserted:
     round 3, a rematerialised ones operand turned the bias sums into garbage);
  B  a non-MFMA instruction (v_accvgpr_read, any VALU / memory instruction) reads the DESTINATION of an asm MFMA inside the
     matrix pipe's write-back window - 11 wait states after a 16x16 MFMA, 19 after a 32x32 one (CDNA3 ISA, XDL write -> VALU
     read; s_nop N counts N + 1) - e.g. an accumulator drain placed too close to the last MFMA;
  C  an asm MFMA reads a VGPR that an LDS read (ds_read* / ds_load*) wrote, with no s_waitcnt lgkmcnt between the two that
     covers that read (lgkmcnt(N) leaves the N youngest LDS operations in flight).

    hipcc ... -S --cuda-device-only wgrad.hip -o /tmp/wgrad.s && python tools/mfma_hazard_check.py /tmp/wgrad.s wgrad_pipe_kernel
Exit code 1 when anything is found.  `tests/test_abi_cpu.py` runs it on every kernel file that issues MFMAs from inline asm."""
import re
import sys


def regs(tok):
    """Register numbers of an operand: VGPRs as n, AGPRs as 1000 + n."""
    tok = tok.strip()
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        base = 1000 if m.group(1) == "a" else 0
        return set(range(base + int(m.group(2)), base + int(m.group(3)) + 1))
    m = re.fullmatch(r"([va])(\d+)", tok)
    return {(1000 if m.group(1) == "a" else 0) + int(m.group(2))} if m else set()


def lgkm_count(s):
    m = re.search(r"lgkmcnt\((\d+)\)", s)
    return int(m.group(1)) if m else None


def check(lines, pattern):
    kernel, in_asm = None, False
    window = []                 # (op, dst regs) of the last instructions (class A)
    live = []                   # [dst regs, wait states left, text] of asm MFMAs still writing back (class B)
    lds = []                    # dst regs of LDS reads in flight, oldest first (class C)
    n, found = 0, []
    for line in lines:
        s = line.strip()
        m = re.match(r"^(_Z\w+):", s)
        if m:
            kernel, window, live, lds, in_asm = m.group(1), [], [], [], False
            continue
        if "#ASMSTART" in s:
            in_asm = True
            continue
        if "#ASMEND" in s:
            in_asm = False
            continue
        if not s or s.startswith(";") or s.startswith(".") or kernel is None or pattern not in kernel:
            continue
        s = s.split(";")[0].strip()
        if not s or s.endswith(":"):
            continue
        op, _, rest = s.partition(" ")
        toks = [t.strip() for t in rest.split(",")] if rest else []
        dst = regs(toks[0]) if toks else set()
        srcs = set()
        for t in toks[1:]:
            srcs |= regs(t.split(" ")[0])
        # ---- wait states this instruction spends
        states = 1
        if op == "s_nop" and toks:
            try:
                states = int(toks[0], 0) + 1
            except ValueError:
                states = 1
        is_mfma = op.startswith("v_mfma")
        # ---- class B: a reader of a destination still inside its window
        if not is_mfma:
            reads = srcs | (dst if op.startswith(("global_store", "buffer_store", "ds_write", "ds_store", "flat_store", "scratch_store")) else set())
            for d, left, text in live:
                if left > 0 and reads & d:
                    found.append(f"B {kernel}: `{s}` reads {fmt(reads & d)} {left} wait state(s) before `{text}` has written it back")
        live = [[d, left - states, text] for d, left, text in live if left - states > 0]
        # ---- class C bookkeeping
        if op == "s_waitcnt":
            c = lgkm_count(s)
            if c is not None:
                lds = lds[len(lds) - c:] if c < len(lds) else lds
                if c == 0:
                    lds = []
        if op.startswith(("ds_read", "ds_load", "ds_bpermute", "ds_permute")):
            lds.append(dst)
        if is_mfma:
            if in_asm:
                n += 1
                src_ab = regs(toks[1]) | regs(toks[2])
                for dist, (pop, pdst) in enumerate(reversed(window[-2:]), 1):
                    if pop.startswith("v_") and not pop.startswith("v_mfma") and pdst & src_ab:
                        found.append(f"A {kernel}: `{pop}` writes {fmt(pdst & src_ab)} {dist} instruction(s) before `{s}`")
                for pend in lds:
                    if pend & (src_ab | regs(toks[3] if len(toks) > 3 else "")):
                        found.append(f"C {kernel}: `{s}` reads {fmt(pend & src_ab)} from an LDS read no lgkmcnt wait covers")
                live.append([dst, 19 if "32x32" in op else 11, s])
        window.append((op, dst))
    return n, found


def fmt(rs):
    return ",".join(("a%d" % (r - 1000)) if r >= 1000 else ("v%d" % r) for r in sorted(rs))


def main(path, pattern):
    n, found = check(open(path).read().split("\n"), pattern)
    for f in found:
        print(f)
    print(f"{n} inline-asm MFMAs checked in kernels matching '{pattern}', {len(found)} hazards")
    return 1 if found else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""))
