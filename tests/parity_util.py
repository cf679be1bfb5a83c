"""Shared tolerance logic of the GPU parity tests.

north_star asks for magnitude bins within 1e-5 relative (float32) of the CPU path.  A float32 FFT cannot meet a
per-bin relative bound on bins that lie far below the frame's peak (a standard float32 CPU FFT -- pocketfft --
shows 1e-3 per-bin relative error on such bins of our synthetic input while staying within 4e-7 of the PEAK), so
the bound is written as

    |P_gpu[k] - P_ref[k]|  <=  REL * P_ref[k]  +  FLOOR * max_k P_ref         REL = 1e-5, FLOOR = 1e-6

(P_ref from the float64 oracle).  For dB columns the same bound is mapped through 10*log10 and DB_SLACK = 3e-5 dB
is added for the float32 rounding of the dB value itself (ulp(110 dB) = 7.6e-6) and the hardware log2 unit.
"""
import numpy as np

REL = 1e-5
FLOOR = 1e-6
DB_SLACK = 3e-5
LOG_FLOOR = np.float64(np.float32(1e-11))


def assert_power_close(p_gpu, p_ref64, what=""):
    p_gpu = np.asarray(p_gpu, dtype=np.float64)
    peak = p_ref64.max(axis=-1, keepdims=True)
    tol = REL * p_ref64 + FLOOR * peak
    err = np.abs(p_gpu - p_ref64)
    bad = err > tol
    assert not bad.any(), f"{what}: {bad.sum()} bins out of tolerance, worst ratio {np.max(err / tol):.3g}"
    strong = p_ref64 > 1e-2 * peak   # bins within 20 dB of the frame peak: plain 1e-5 relative must hold
    return float(np.max(err[strong] / p_ref64[strong])) if strong.any() else 0.0


def assert_db_close(db_gpu, db_ref, p_ref64, what="", peak=None):
    """db_ref: oracle dB (float32); p_ref64: the (mixed) linear power the oracle took the log of.  peak: the frame peak the
    FLOOR term refers to; default: the peak of the mixed column.  A selecting mix (Min, Max, Left, Right) picks single
    channels' bins, whose float32 error scales with THAT channel's frame peak: such callers pass the largest per-channel
    peak of the frame (Min over 8 noise channels otherwise shrinks the yardstick to the weakest channel's bins)."""
    if peak is None:
        peak = p_ref64.max(axis=-1, keepdims=True)
    r = REL + FLOOR * peak / (p_ref64 + LOG_FLOOR)
    tol = 10.0 * np.log10(1.0 + r) + DB_SLACK
    err = np.abs(db_gpu.astype(np.float64) - db_ref.astype(np.float64))
    bad = err > tol
    assert not bad.any(), f"{what}: {bad.sum()} dB values out of tolerance, worst ratio {np.max(err / tol):.3g}, max err {err.max():.3g} dB"
    return float(err.max())


def mixed_power_f64(oracle, samples, n, hop, feedblocks, win, mode, power_scale=1.0):
    """Float64 view of the power the oracle feeds to 10*log10 (mix done like the reference, in float32)."""
    pw = oracle.stft_db_reference(samples, n, hop, feedblocks, win, return_power=True, power_scale=power_scale)
    return oracle.mix_channels(pw.astype(np.float32), mode).astype(np.float64)
