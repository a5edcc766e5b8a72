"""TEST INFRASTRUCTURE ONLY — CPU oracle for the U-Net denoising train step.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and only as the checker (never as the thing measured or shipped).  The
product path (``pokemon_sprite_generator_amd``) never imports this package and
fails loudly when its HIP extension is missing.
"""
