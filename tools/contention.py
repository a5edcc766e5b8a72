"""CU-contention rehearsal of the data-parallel step on ONE GPU: the batch-256 bf16 train step on a HIP stream
restricted to 256 / 240 / 224 / 192 CUs (hipExtStreamCreateWithCUMask) - the CUs an overlapped RCCL all-reduce's channels would
occupy during backward - with the tile choosers planning for all 256 CUs ("blind") and for the CUs they really have
("matched", psg_set_available_cus).  Prints ms/step and per-family ms.   python tools/contention.py [batch]"""
import ctypes as C
import json
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pokemon_sprite_generator_amd as psg
from pokemon_sprite_generator_amd import _lib

KINDS = ["conv fwd", "conv dgrad", "wgrad", "attention", "groupnorm"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lib = _lib.init(0)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
unet = psg.UNet(compute_dtype=torch.bfloat16).to(dev)
st = psg.DiffusionStepper(unet, psg.NoiseScheduler(), distributed=False)
lat, txt = torch.randn(B, 8, 27, 27, device=dev) * 1.2, torch.randn(B, 32, 256, device=dev)
t = torch.randint(0, 1000, (B,), device=dev)
rows = []
for ncu in (256, 240, 224, 192):
    sp = C.c_void_p()
    _lib.check(lib.psg_stream_create_cu_mask(ncu, C.byref(sp)), "psg_stream_create_cu_mask")
    ext = torch.cuda.ExternalStream(sp.value, device=dev)
    for plan in (("blind", 256, 0), ("matched", ncu, 0), ("matched, <= 3 rounds", ncu, 3)) if ncu != 256 else (("blind", 256, 0), ("plan 224", 224, 0), ("plan 224, <= 3 rounds", 224, 3)):
        _lib.set_available_cus(0, 0 if plan[1] == 256 else plan[1], plan[2])
        ext.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(ext):
            for _ in range(2):
                st.train_step(lat, txt, t)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                st.train_step(lat, txt, t)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 4
            _lib.check(lib.psg_profile_begin(), "begin")
            for _ in range(2):
                st.train_step(lat, txt, t)
            torch.cuda.synchronize()
            n = len(KINDS)
            fm, wk, cnt = (C.c_double * n)(), (C.c_double * n)(), (C.c_int64 * n)()
            _lib.check(lib.psg_profile_end(fm, wk, cnt, n), "end")
        row = {"cus": ncu, "plan": plan[0], "ms_per_step": round(ms, 2), **{k: round(fm[i] / 2, 2) for i, k in enumerate(KINDS)}}
        rows.append(row)
        print(json.dumps(row), flush=True)
    torch.cuda.current_stream(dev).wait_stream(ext)
    torch.cuda.synchronize()
    _lib.set_available_cus(0, 0)
    del ext
    _lib.check(lib.psg_stream_destroy(sp), "psg_stream_destroy")
base = rows[0]["ms_per_step"]
print("\n| CUs | plan | ms/step | vs 256 | " + " | ".join(KINDS) + " |\n|---|---|---|---|" + "---|" * len(KINDS))
for r in rows:
    print(f"| {r['cus']} | {r['plan']} | {r['ms_per_step']} | {r['ms_per_step'] / base:.3f} | " + " | ".join(str(r[k]) for k in KINDS) + " |")
