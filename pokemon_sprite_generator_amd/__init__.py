"""pokemon_sprite_generator_amd — MI355X-native U-Net denoising train step.

Drop-in for the hot path of GabrieleConte/pokemon-sprite-generator
(src/models/unet.py + src/training/improved_diffusion_trainer.py): the same
Python surface, computed by hand-written gfx950 HIP kernels in libpsg_hip.so
(C ABI: include/psg_hip.h).  Importing the package never touches the GPU; using
any op without the built library raises (no CPU fallback).
"""
from ._lib import LIB_PATH, PsgError
from .scheduler import NoiseScheduler
from .unet import CrossAttentionBlock, ResBlock, TimestepEmbedding, UNet, UNetBlock
from .optim import FusedAdamW, GradArena, ParamArena
from .ddp import BucketedAllReduce
from .trainer import DiffusionStepper, DiffusionTrainer, ImprovedDiffusionTrainer
from .inference import LatentGenerator, LinearNoiseScheduler, gradio_ddpm_sample
from .vae import PokemonVAE, VAEDecoder, VAEEncoder

__all__ = ["UNet", "UNetBlock", "ResBlock", "CrossAttentionBlock", "TimestepEmbedding", "NoiseScheduler",
           "ImprovedDiffusionTrainer", "DiffusionTrainer", "DiffusionStepper", "FusedAdamW", "GradArena", "ParamArena",
           "BucketedAllReduce", "LatentGenerator", "LinearNoiseScheduler", "gradio_ddpm_sample", "PokemonVAE", "VAEEncoder", "VAEDecoder", "PsgError", "LIB_PATH"]
__version__ = "0.1.0"
