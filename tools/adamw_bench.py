"""AdamW kernel bandwidth on a 640 M-element flat arena: float4 path, float4 + bf16 shadow, scalar path."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pokemon_sprite_generator_amd import _lib
from pokemon_sprite_generator_amd._lib import ptr, stream_ptr, check
lib = _lib.init(0)
n = 640_000_000
p, g, m, v = (torch.randn(n + 8, device='cuda') * 0.01 for _ in range(4))
v.abs_()
sh = torch.empty(n + 8, dtype=torch.bfloat16, device='cuda')
nsq = torch.ones(1, device='cuda')
def run(off, shadow, tag):
    args = lambda: (ptr(p[off:]), ptr(g[off:]), ptr(m[off:]), ptr(v[off:]), n, 1e-4, 0.9, 0.999, 1e-6, 0.01, 3, ptr(nsq), 1.0, None,
                    ptr(sh) if shadow else None, stream_ptr())
    check(lib.psg_adamw_f32(*args()), "adamw")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        check(lib.psg_adamw_f32(*args()), "adamw")
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    gb = n * (28 + (2 if shadow else 0)) / 1e9
    print(f"{tag:18s} {ms:6.3f} ms  {gb / ms:6.2f} TB/s")
run(0, False, "float4")
run(0, True, "float4+shadow")
run(1, False, "scalar (unaligned)")
