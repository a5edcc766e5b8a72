"""Static check of a hipcc -S dump: an inline-asm MFMA is invisible to the compiler's hazard recogniser, so a VALU instruction
that writes one of its source VGPRs within the two preceding instructions gets no wait states.  Lists every such place.
    hipcc ... -S --cuda-device-only wgrad.hip -o /tmp/wgrad.s && python tools/mfma_hazard_check.py /tmp/wgrad.s wgrad_pipe_kernel"""
import re, sys


def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def main(path, pattern):
    kernel, window, bad, n = None, [], 0, 0
    for line in open(path):
        s = line.strip()
        m = re.match(r"^(_Z\w+):", s)
        if m:
            kernel, window = m.group(1), []
            continue
        if not s or s.startswith(";") or s.startswith(".") or kernel is None or pattern not in kernel:
            continue
        op, _, rest = s.partition(" ")
        toks = [t.strip() for t in rest.split(",")]
        if op.startswith("v_mfma"):
            n += 1
            src = regs(toks[1]) | regs(toks[2])
            for dist, (pop, pdst) in enumerate(reversed(window[-2:]), 1):
                if pop.startswith("v_") and not pop.startswith("v_mfma") and pdst & src:
                    bad += 1
                    print(f"{kernel}: {pop} writes v{sorted(pdst & src)} {dist} instruction(s) before: {s}")
        window.append((op, regs(toks[0]) if toks else set()))
    print(f"{n} MFMAs checked in kernels matching '{pattern}', {bad} hazards")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""))
