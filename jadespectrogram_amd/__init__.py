"""jadespectrogram_amd -- MI355X-native (gfx950) STFT spectrogram engine behind the reference's Spectrogram API.

The compute path is hand-written HIP in libjsg.so (csrc/), reached through the C-ABI of include/jsg.h.
Importing this package does not load the library; the first call does, and fails loudly if it is missing.
"""
from . import capi  # noqa: F401
from .capi import JsgError  # noqa: F401
from .spectrogram import (CColorPalette, Plan, Spectrogram, SpectrogramDisplay, StftLaunch, colormap, colormap_lut,  # noqa: F401
                          colormap_range, columns_from_tail_layout, feed_samples, memsize_blocks, next_power_of_2, stft_db, stft_db_batches, stft_db_strided, stft_db_strided_kernel_name, stft_image, stft_image_needs_scratch, stft_image_strided, stft_image_strided_needs_scratch, stft_kernel_name, window)

__all__ = ["Spectrogram", "SpectrogramDisplay", "CColorPalette", "Plan", "stft_db", "colormap", "window", "colormap_lut",
           "colormap_range", "feed_samples", "memsize_blocks", "next_power_of_2", "JsgError", "capi"]
