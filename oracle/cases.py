"""TEST INFRASTRUCTURE ONLY — the parity cases shared by make_golden.py and tests/.

Each case names the reference class (src/models/unet.py) it exercises, its
constructor arguments and the hash-generator seeds for weights and inputs, so
fixtures hold only OUTPUTS: inputs and weights are regenerated bit-identically
from oracle/hashgen.py on any machine.
"""

# (name, cin, cout, H=W, batch) — ResBlock(in,out,time_emb_dim=128,text_emb_dim=256) unet.py:63
RESBLOCK_CASES = [
    ("res_64_128_h7", 64, 128, 7, 2),     # with 1x1 skip conv (unet.py:95-96)
    ("res_64_64_h5", 64, 64, 5, 3),       # identity skip (unet.py:98)
    ("res_128_64_h4", 128, 64, 4, 2),     # concat-style 2C->C, 4x4 level
]

# (name, channels, heads, H=W, batch, seq) — CrossAttentionBlock(channels,text_dim=256,heads) unet.py:140
ATTN_CASES = [
    ("attn_128_h8_l16", 128, 8, 4, 2, 32),
    ("attn_128_h4_l49", 128, 4, 7, 2, 32),
    ("attn_640_h8_l196", 640, 8, 14, 1, 32),   # real level-1 shape: L=196, d=80
    ("attn_1280_h8_l16_s20", 1280, 8, 4, 1, 20),  # d=160, ragged S (tokenizer pads to longest)
]

TIME_EMBED_T = [0, 1, 500, 999, 37, 250]

# full-width U-Net cases: (name, weight mode, batch, timesteps, heads)
UNET_CASES = [
    ("unet_default_b1", "default", 1, [500], 8),
    ("unet_stress_b1", "stress", 1, [500], 8),
    ("unet_stress_b2", "stress", 2, [500, 3], 8),
    ("unet_stress_b2_h4", "stress", 2, [999, 0], 4),   # CLI path really runs 4 heads
]

WEIGHT_SEED = 20250808
INPUT_SEED = 1234
TRAIN_CASE = ("train_stress_b2", "stress", 2, [500, 3], 8)
SAMPLE_CASE = ("sample_stress_fast", "stress", 1, 8)
