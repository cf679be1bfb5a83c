"""Shared tolerance logic of the GPU parity tests.

north_star asks for magnitude bins within 1e-5 relative (float32) of the CPU path.  A float32 FFT cannot meet a
per-bin relative bound on bins that lie far below the frame's peak (a standard float32 CPU FFT -- pocketfft --
shows 1e-3 per-bin relative error on such bins of our synthetic input while staying within 4e-7 of the PEAK), so
the bound is written as

    |P_gpu[k] - P_ref[k]|  <=  REL * P_ref[k]  +  FLOOR(n) * max_k P_ref         REL = 1e-5

(P_ref from the float64 oracle).  FLOOR is a REGRESSION GUARD, not slack: per FFT size it sits at about 1.4 x the worst
error relative to the frame peak that the kernels show over tools/accuracy_report.py's sweep (all six windows, 1 / 2 / 3 / 8
channels, 50 % and 75 % overlap, sine + noise / noise / full-scale sine / chirp, both kernels of 2048 and 4096 points; MI355X,
round 3):  512: 4.5e-7, 1024: 4.9e-7, 2048: 5.3e-7 (B: 5.0e-7), 4096: 5.2e-7 (B: 7.3e-7), 8192: 6.0e-7.  A kernel change that
loses half a bit fails the suite.  Bins within 20 dB of the frame peak must also hold STRONG_REL = 5e-6 plain relative
(measured worst: 3.0e-6), twice as tight as north_star's 1e-5.
For dB columns the same bound is mapped through 10*log10 and DB_SLACK = 3e-5 dB is added for the float32 rounding of the
dB value itself (ulp(110 dB) = 7.6e-6) and the hardware log2 unit.
"""
import numpy as np

REL = 1e-5
STRONG_REL = 5e-6
FLOOR_BY_N = {512: 6.5e-7, 1024: 7.0e-7, 2048: 7.5e-7, 4096: 1.0e-6, 8192: 8.5e-7}
FLOOR = 1.0e-6          # (FFT size unknown to the caller: the loosest of the table)
DB_SLACK = 3e-5
LOG_FLOOR = np.float64(np.float32(1e-11))


def floor_for(n_bins):
    """FLOOR for a column of n_bins = n/2+1 values."""
    return FLOOR_BY_N.get(2 * (int(n_bins) - 1), FLOOR)


def assert_power_close(p_gpu, p_ref64, what=""):
    p_gpu = np.asarray(p_gpu, dtype=np.float64)
    peak = p_ref64.max(axis=-1, keepdims=True)
    tol = REL * p_ref64 + floor_for(p_ref64.shape[-1]) * peak
    err = np.abs(p_gpu - p_ref64)
    bad = err > tol
    assert not bad.any(), f"{what}: {bad.sum()} bins out of tolerance, worst ratio {np.max(err / tol):.3g}"
    strong = p_ref64 > 1e-2 * peak   # bins within 20 dB of the frame peak: plain relative, tighter than north_star's 1e-5
    worst = float(np.max(err[strong] / p_ref64[strong])) if strong.any() else 0.0
    assert worst <= STRONG_REL, f"{what}: relative error {worst:.3g} on a bin within 20 dB of the frame peak"
    return worst


def assert_db_close(db_gpu, db_ref, p_ref64, what="", peak=None):
    """db_ref: oracle dB (float32); p_ref64: the (mixed) linear power the oracle took the log of.  peak: the frame peak the
    FLOOR term refers to; default: the peak of the mixed column.  A selecting mix (Min, Max, Left, Right) picks single
    channels' bins, whose float32 error scales with THAT channel's frame peak: such callers pass the largest per-channel
    peak of the frame (Min over 8 noise channels otherwise shrinks the yardstick to the weakest channel's bins)."""
    if peak is None:
        peak = p_ref64.max(axis=-1, keepdims=True)
    r = REL + floor_for(p_ref64.shape[-1]) * peak / (p_ref64 + LOG_FLOOR)
    tol = 10.0 * np.log10(1.0 + r) + DB_SLACK
    err = np.abs(db_gpu.astype(np.float64) - db_ref.astype(np.float64))
    bad = err > tol
    assert not bad.any(), f"{what}: {bad.sum()} dB values out of tolerance, worst ratio {np.max(err / tol):.3g}, max err {err.max():.3g} dB"
    return float(err.max())


def mixed_power_f64(oracle, samples, n, hop, feedblocks, win, mode, power_scale=1.0):
    """Float64 view of the power the oracle feeds to 10*log10 (mix done like the reference, in float32)."""
    pw = oracle.stft_db_reference(samples, n, hop, feedblocks, win, return_power=True, power_scale=power_scale)
    return oracle.mix_channels(pw.astype(np.float32), mode).astype(np.float64)
