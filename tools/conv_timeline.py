"""Per-workgroup phase timestamps of one conv launch (needs the PSG_ABL=8 build: PSG_LIB_PATH=build_abl/a8/libpsg_hip.so).
Phases: start -> first tile landed -> K loop done -> stores issued -> stores complete (10 ns ticks of s_memrealtime)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pokemon_sprite_generator_amd import ops, _lib
lib = _lib.init(0)
B = 256
dbg = torch.zeros(8 * 65536, dtype=torch.int64, device="cuda")
for H, Cin, Cout, ks in [(14, 640, 640, 1), (14, 640, 1920, 1), (7, 1280, 1280, 1), (14, 640, 640, 3)]:
    x = torch.randn(B, H, H, Cin, device="cuda").bfloat16()
    w = (torch.randn(Cout, Cin, ks, ks, device="cuda") * 0.02)
    b = torch.randn(Cout, device="cuda")
    with torch.no_grad():
        for _ in range(3):
            ops.conv2d(x, w, b)
        torch.cuda.synchronize()
        dbg.zero_(); torch.cuda.synchronize()
        os.environ["PSG_DBG_PTR"] = hex(dbg.data_ptr())
        ops.conv2d(x, w, b)
        torch.cuda.synchronize()
        del os.environ["PSG_DBG_PTR"]
    d = dbg.cpu().numpy().reshape(-1, 8)
    d = d[d[:, 0] != 0]
    n = len(d)
    t0 = d[:, 0].min()
    ph = np.stack([d[:, 1] - d[:, 0], d[:, 2] - d[:, 1], d[:, 3] - d[:, 2], d[:, 4] - d[:, 3], d[:, 4] - d[:, 0]], 1) * 0.01
    print(f"{H}x{H} {Cin}->{Cout} k{ks}: {n} workgroups, launch span {(d[:, 4].max() - t0) * 0.01:.1f} us")
    print("   mean us: first tile %.2f | K loop %.2f | epilogue issue %.2f | store drain %.2f | total %.2f" % tuple(ph.mean(0)))
    print("   epilogue: staging (acc -> LDS) %.2f us, flush (LDS -> global stores issued) %.2f us" % (((d[:, 7] - d[:, 2]) * 0.01).mean(), ((d[:, 3] - d[:, 7]) * 0.01).mean()))
    print("   p90  us: first tile %.2f | K loop %.2f | epilogue issue %.2f | store drain %.2f | total %.2f" % tuple(np.percentile(ph, 90, 0)))
    # one CU's timeline: group by (xcd = bid & 7, hw_id cu/sh/se bits)
    bid = np.arange(len(dbg) // 8)[(dbg.cpu().numpy().reshape(-1, 8)[:, 0] != 0)]
    key = (bid & 7) * 65536 + (d[:, 5] & 0xFF00)
    k0 = key[0]
    sel = np.where(key == k0)[0]
    sel = sel[np.argsort(d[sel, 0])]
    print("   CU of workgroup 0 ran %d workgroups: (lds_base, start, tile0, kdone, issued, drained) us" % len(sel))
    for i in sel[:10]:
        print("     bid %5d lds %3d  %7.2f %7.2f %7.2f %7.2f %7.2f" % (bid[i], d[i, 6], *[(d[i, k] - t0) * 0.01 for k in range(5)]))
