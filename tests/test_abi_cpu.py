"""CPU suite: the C-ABI library builds for gfx950, loads, and exports every symbol include/psg_hip.h
declares; argument validation returns error codes (no compute is launched without a GPU)."""
import ctypes as C
import os
import sys
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from pokemon_sprite_generator_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    return _lib.load()


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "psg_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(psg_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from pokemon_sprite_generator_amd import _lib
    names = _declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"libpsg_hip.so does not export {n}"
        assert n in _lib.SIGNATURES, f"{n} declared in the header but not bound in _lib.SIGNATURES"
    for n in _lib.SIGNATURES:
        assert n in names, f"{n} bound but not declared in include/psg_hip.h"
    assert lib.psg_version() >= 100


def test_struct_layouts_match_header():
    """Compile the header with gcc and compare sizeof/offsetof with the ctypes mirrors."""
    import subprocess
    import tempfile
    from pokemon_sprite_generator_amd._lib import ConvDesc, WgradDesc
    fields_c = [f for f, _ in ConvDesc._fields_]
    fields_w = [f for f, _ in WgradDesc._fields_]
    prog = '#include <stdio.h>\n#include <stddef.h>\n#include "psg_hip.h"\nint main(void){\n'
    prog += 'printf("%zu\\n", sizeof(psg_conv_desc));\n' + "".join(f'printf("%zu\\n", offsetof(psg_conv_desc, {f}));\n' for f in fields_c)
    prog += 'printf("%zu\\n", sizeof(psg_wgrad_desc));\n' + "".join(f'printf("%zu\\n", offsetof(psg_wgrad_desc, {f}));\n' for f in fields_w)
    prog += "return 0;}\n"
    with tempfile.TemporaryDirectory() as td:
        src, exe = os.path.join(td, "abi.c"), os.path.join(td, "abi")
        open(src, "w").write(prog)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        vals = [int(v) for v in subprocess.check_output([exe]).split()]
    want = [C.sizeof(ConvDesc)] + [getattr(ConvDesc, f).offset for f in fields_c]
    want += [C.sizeof(WgradDesc)] + [getattr(WgradDesc, f).offset for f in fields_w]
    assert vals == want


def test_argument_validation_without_gpu(lib):
    """Bad arguments are rejected on the host before any launch: error code + message."""
    from pokemon_sprite_generator_amd._lib import ConvDesc, WgradDesc
    d = ConvDesc()
    assert lib.psg_conv_fwd(C.byref(d), None) < 0
    assert b"null" in lib.psg_last_error()
    d.x = d.w = d.y = 0x1000
    d.dtype = 7
    assert lib.psg_conv_fwd(C.byref(d), None) == -2                           # PSG_ERR_DTYPE
    d.dtype = 1
    d.B, d.Hi, d.Wi, d.Cin, d.Ho, d.Wo, d.Cout = 2, 7, 7, 12, 7, 7, 64       # Cin not a multiple of 8 for bf16
    d.ksize, d.stride, d.pad = 3, 1, 1
    d.ldx, d.ldy = 12, 64
    assert lib.psg_conv_fwd(C.byref(d), None) == -1                           # PSG_ERR_SHAPE
    assert b"Cin" in lib.psg_last_error()
    d.Cin, d.ldx, d.Ho = 16, 16, 9                                            # inconsistent geometry
    assert lib.psg_conv_fwd(C.byref(d), None) == -1
    w = WgradDesc()
    assert lib.psg_conv_wgrad_workspace_bytes(C.byref(w)) == -1
    assert lib.psg_groupnorm_fwd(None, 0, None, 0, None, None, None, None, 1, 1, 32, 32, 1e-5, 0, 0, None, None) == -6
    assert lib.psg_kpad(72, 1) == 128 and lib.psg_kpad(72, 0) == 96 and lib.psg_kpad(2880, 1) == 2880
    assert lib.psg_attn_fwd(0x10, 8, 0x10, 8, 0x10, 8, 0x10, 8, 0x10, 1, 1, 4, 4, 6, 1.0, 0.0, 0, 0, None) == -1   # d % 4


def test_product_has_no_cpu_fallback():
    """Ops refuse CPU tensors loudly instead of rerouting to an eager/oracle path."""
    import pokemon_sprite_generator_amd as psg
    from pokemon_sprite_generator_amd import ops
    with pytest.raises(psg.PsgError):
        ops.group_norm(torch.zeros(1, 2, 2, 32), torch.ones(32), torch.zeros(32), 32)
    with pytest.raises(psg.PsgError):
        psg.NoiseScheduler().add_noise(torch.zeros(1, 8, 2, 2), torch.zeros(1, 8, 2, 2), torch.zeros(1, dtype=torch.long))
    src = ""
    pkg = os.path.join(ROOT, "pokemon_sprite_generator_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            src += open(os.path.join(pkg, f)).read()
    assert "import oracle" not in src and "from oracle" not in src, "the product must never import the oracle"


def test_inline_asm_mfmas_have_no_valu_write_hazard(tmp_path):
    """wgrad_pipe_kernel issues its MFMAs as inline asm (accumulators pinned to AGPRs), which hides them from hipcc's hazard
    recogniser: a VALU write of a source VGPR directly in front of one gets no wait states (round 3: a rematerialised ones
    operand turned the bias sums into garbage).  tools/mfma_hazard_check.py scans the ISA of the shipped build flags for it."""
    import subprocess, sys
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    asm = str(tmp_path / "wgrad.s")
    src = os.path.join(ROOT, "pokemon_sprite_generator_amd", "csrc", "wgrad.hip")
    subprocess.run([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-S", "--cuda-device-only", src, "-o", asm],
                   check=True, capture_output=True, timeout=600)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "mfma_hazard_check.py"), asm, "wgrad_pipe_kernel"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert " 0 hazards" in r.stdout and not r.stdout.startswith("0 inline-asm MFMAs"), r.stdout


def test_mfma_hazard_checker_finds_planted_hazards():
    """The checker's three classes on synthetic listings: A - VALU write of a source right in front of an
    asm MFMA; B - a read of the MFMA's destination inside the write-back window (the accumulator drain too close); C - a source
    that an LDS read wrote with no lgkmcnt wait in between.  And the clean forms of each pass."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import mfma_hazard_check as H

    def run(body):
        return H.check(("_Z6kernelv:\n" + body).split("\n"), "kernel")

    mf = "\t;;#ASMSTART\n\tv_mfma_f32_16x16x32_bf16 a[0:3], v[10:13], v[20:23], a[0:3]\n\t;;#ASMEND\n"
    n, f = run("\tv_mov_b32_e32 v10, 1.0\n" + mf)
    assert n == 1 and len(f) == 1 and f[0].startswith("A "), f
    n, f = run("\tv_mov_b32_e32 v10, 1.0\n\ts_nop 1\n\tv_add_u32_e32 v1, v2, v3\n" + mf)
    assert n == 1 and not f, f
    n, f = run(mf + "\ts_nop 3\n\tv_accvgpr_read_b32 v5, a2\n")
    assert len(f) == 1 and f[0].startswith("B "), f
    n, f = run(mf + "\ts_nop 15\n\tv_accvgpr_read_b32 v5, a2\n")
    assert not f, f
    n, f = run(mf + "\tv_mfma_f32_16x16x32_bf16 a[0:3], v[10:13], v[20:23], a[0:3]\n")        # back-to-back accumulation: interlocked
    assert not f, f
    n, f = run("\tds_read_b128 v[10:13], v40\n" + mf)
    assert len(f) == 1 and f[0].startswith("C "), f
    n, f = run("\tds_read_b128 v[10:13], v40\n\tds_read_b128 v[30:33], v41\n\ts_waitcnt lgkmcnt(1)\n" + mf)
    assert not f, f
    n, f = run("\tds_read_b128 v[30:33], v41\n\tds_read_b128 v[10:13], v40\n\ts_waitcnt lgkmcnt(1)\n" + mf)
    assert len(f) == 1 and f[0].startswith("C "), f
    # an MFMA the COMPILER issued (outside an asm block) is its own business: not counted, not flagged
    n, f = run("\tv_mov_b32_e32 v10, 1.0\n\tv_mfma_f32_16x16x32_bf16 v[0:3], v[10:13], v[20:23], v[0:3]\n")
    assert n == 0 and not f


def test_workspace_pool_never_frees_a_buffer_a_graph_holds():
    """a captured hipGraph points into the scratch buffer of its capture stream, and torch recycles stream handles.
    The pool's rules (pokemon_sprite_generator_amd/_lib.py::WorkspacePool), on a fake allocator: a held buffer survives a later,
    larger request on the same key (the key gets a NEW buffer); growth inside a capture raises; drop() releases the entry."""
    from pokemon_sprite_generator_amd._lib import PsgError, WorkspacePool

    class Buf:
        live = 0

        def __init__(self, n):
            self.n = n
            Buf.live += 1

        def numel(self):
            return self.n

        def __del__(self):
            Buf.live -= 1

    pool = WorkspacePool(lambda n: Buf(n))
    key = (0, 1234)
    a = pool.get(key, 10)
    assert pool.get(key, 5) is a and a.numel() == 1 << 20          # grow-only, 1 MiB floor
    handle = pool.hold(key)                                        # a graph captured on this stream points into `a`
    pool.frozen = True
    with pytest.raises(PsgError):
        pool.get(key, 2 << 20)                                     # growth inside a capture
    pool.frozen = False
    b = pool.get(key, 2 << 20)                                     # a recycled stream handle, a larger eager request
    assert b is not a and handle[1] is a
    del a
    import gc
    gc.collect()
    assert Buf.live == 2                                           # the held buffer is still alive (the graph can replay)
    pool.drop(handle)
    del handle
    gc.collect()
    assert Buf.live == 1 and len(pool) == 1                        # only the current buffer remains
    h2 = pool.hold(key)
    pool.drop(h2)                                                  # owner of the CURRENT buffer closed: the entry goes too
    del h2, b
    gc.collect()
    assert len(pool) == 0 and Buf.live == 0
    with pytest.raises(PsgError):
        pool.hold((0, 99))                                         # nothing was sized on that stream
