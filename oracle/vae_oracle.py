"""TEST INFRASTRUCTURE ONLY — CPU restatement of the frozen VAE either side of the U-Net path,
plain torch functional ops on a state_dict, every function citing the reference lines it follows
(src/models/vae_decoder.py).  Pinned to the reference by oracle/make_golden_vae.py (runs the reference module itself).
Never imported by the product."""
import math

import torch
import torch.nn.functional as F


def resnet_block(x, sd, p, groups=32):
    """vae_decoder.py:24-32."""
    h = F.silu(F.group_norm(x, groups, sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5))
    h = F.conv2d(h, sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1)
    h = F.silu(F.group_norm(h, groups, sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5))
    h = F.conv2d(h, sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1)
    if p + "shortcut.weight" in sd:
        x = F.conv2d(x, sd[p + "shortcut.weight"], sd[p + "shortcut.bias"])
    return h + x


def cross_attention_block(x, text, sd, p, num_heads=8):
    """vae_decoder.py:49-65.  NOTE :56-57: k / v are [b, S, C] tensors RESHAPED (not transposed) to
    [b, heads, head_dim, S] - the reference's own layout, reproduced."""
    b, c, h, w = x.shape
    hd = c // num_heads
    xn = F.group_norm(x, 32, sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-5)
    q = F.conv2d(xn, sd[p + "q.weight"], sd[p + "q.bias"]).reshape(b, num_heads, hd, h * w)
    k = F.linear(text, sd[p + "k.weight"], sd[p + "k.bias"]).reshape(b, num_heads, hd, -1)
    v = F.linear(text, sd[p + "v.weight"], sd[p + "v.bias"]).reshape(b, num_heads, hd, -1)
    attn = torch.softmax(q.transpose(-2, -1) @ k / math.sqrt(hd), dim=-1)
    o = (attn @ v.transpose(-2, -1)).transpose(-2, -1).reshape(b, c, h, w)
    return F.conv2d(o, sd[p + "proj.weight"], sd[p + "proj.bias"]) + x


def vae_encode(sd, x, eps):
    """VAEEncoder.forward, vae_decoder.py:104-125; `eps` is the randn_like draw of :121."""
    h = x
    pads = {0: 1, 3: 1, 6: 2}
    for i in range(13):
        p = f"encoder.{i}."
        if i in pads:
            h = F.relu(F.conv2d(h, sd[p + "weight"], sd[p + "bias"], stride=2, padding=pads[i]))       # conv + the ReLU at i+1
        elif p + "norm1.weight" in sd:
            h = resnet_block(h, sd, p)
    mu = F.conv2d(h, sd["mu_proj.weight"], sd["mu_proj.bias"], padding=1)
    logvar = F.conv2d(h, sd["logvar_proj.weight"], sd["logvar_proj.bias"], padding=1)
    return mu + eps * torch.exp(0.5 * logvar), mu, logvar


def vae_decode(sd, latent, text):
    """VAEDecoder.forward, vae_decoder.py:177-222."""
    x = F.conv2d(latent, sd["latent_proj.weight"], sd["latent_proj.bias"], padding=1)
    for i in range(1, 6):
        x = resnet_block(x, sd, f"block{i}_resnet1.")
        x = cross_attention_block(x, text, sd, f"block{i}_attn.")
        x = resnet_block(x, sd, f"block{i}_resnet2.")
        if i in (2, 3):
            x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
        elif i == 4:
            x = F.interpolate(x, size=(215, 215), mode="bilinear", align_corners=False)
    x = F.silu(F.group_norm(x, 8, sd["final_conv.0.weight"], sd["final_conv.0.bias"], 1e-5))
    return torch.tanh(F.conv2d(x, sd["final_conv.2.weight"], sd["final_conv.2.bias"], padding=1))
