"""shot-vae_amd: MI355X-native (gfx950 HIP) implementation of the SHOT-VAE training hot path behind
the reference's Python API (FengHZ/SHOT-VAE: shot_vae_model/vae.py, lib/criterion.py,
lib/utils/mixup.py, the step of main_shot_vae.py)."""
