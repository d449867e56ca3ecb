"""MI355X-native LDP denoising hot path (planner U-Net DDPM/DDIM loop, IDM loop, StableVAE encode)."""
__version__ = "0.1.0"
