/* TEST INFRASTRUCTURE -- never linked into or loaded by the product (jadespectrogram_amd/).
 *
 * CPU mirror of the GPU kernel's float32 arithmetic: the window multiply, the N/2-point complex FFT in the kernel's own
 * factorisation and operation order (radix butterflies, twiddle tables, fused multiply-adds exactly where the kernel's packed
 * instructions fuse, including the window multiply that is folded into the first butterfly layer), the paired real-split post pass, |X|^2, the channel mix and -- optionally -- the dB value through the shared
 * float32 logarithm (jadespectrogram_amd/csrc/jsg_exact_math.h, included from the product tree so that both sides compile the SAME
 * source line by line).  IEEE 754 add / multiply / fma are defined bit for bit, the tables are built by the same double-precision
 * expressions (csrc/jsg_stft_kernel.h: fill_tables), so the mirror's linear power equals the GPU's in every bit, for every plan:
 * tests/test_gpu_mirror.py.  What it restates: stft_db_kernel of csrc/jsg_stft_kernel.h, which replaces Spectrogram.cpp:50-119 +
 * :137-145 + spectrum::power (call site :144) of the reference.  The float64 oracle (jsg_oracle.py) stays the accuracy yardstick; this
 * file is the bit-exactness yardstick (build-plan steps 2-3).
 *
 * Compile with -ffp-contract=off (oracle/Makefile): nothing may be fused that is not written as fmaf here.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../jadespectrogram_amd/csrc/jsg_exact_math.h"

typedef struct { float x, y; } cf;

static inline cf cf_add(cf a, cf b) { cf r = {a.x + b.x, a.y + b.y}; return r; }
static inline cf cf_sub(cf a, cf b) { cf r = {a.x - b.x, a.y - b.y}; return r; }
static inline cf cf_scale(cf a, float s) { cf r = {a.x * s, a.y * s}; return r; }
/* a + (-i) b, a - (-i) b   (jsg_stft_kernel.h: add_mi / sub_mi) */
static inline cf add_mi(cf a, cf b) { cf r = {a.x + b.y, a.y - b.x}; return r; }
static inline cf sub_mi(cf a, cf b) { cf r = {a.x - b.y, a.y + b.x}; return r; }
/* a + conj(b), a - conj(b) */
static inline cf add_conj(cf a, cf b) { cf r = {a.x + b.x, a.y - b.y}; return r; }
static inline cf sub_conj(cf a, cf b) { cf r = {a.x - b.x, a.y + b.y}; return r; }
/* a * w: t = (a.x w.x, a.y w.x); r = (fma(a.y, -w.y, t.x), fma(a.x, w.y, t.y))   (cmul: v_pk_mul_f32 + v_pk_fma_f32) */
static inline cf cmul(cf a, cf w) {
    const float tx = a.x * w.x, ty = a.y * w.x;
    cf r = {fmaf(a.y, -w.y, tx), fmaf(a.x, w.y, ty)};
    return r;
}

static const float kCos32[8] = {1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f,
                                0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508977173f, 0.19509032201612826785f};
static const float kSin32[8] = {0.0f, 0.19509032201612826785f, 0.38268343236508977173f, 0.55557023301960222474f,
                                0.70710678118654752440f, 0.83146961230254523708f, 0.92387953251128675613f, 0.98078528040323044913f};
static const float kCos64[16] = {1.00000000000000000000f, 0.99518472667219692873f, 0.98078528040323043058f, 0.95694033573220882438f, 0.92387953251128673848f, 0.88192126434835504956f, 0.83146961230254523567f, 0.77301045336273699338f, 0.70710678118654757274f, 0.63439328416364548779f, 0.55557023301960228867f, 0.47139673682599780857f, 0.38268343236508983729f, 0.29028467725446233105f, 0.19509032201612833135f, 0.09801714032956077016f};
static const float kSin64[16] = {0.00000000000000000000f, 0.09801714032956060363f, 0.19509032201612824808f, 0.29028467725446233105f, 0.38268343236508978178f, 0.47139673682599764204f, 0.55557023301960217765f, 0.63439328416364548779f, 0.70710678118654746172f, 0.77301045336273699338f, 0.83146961230254523567f, 0.88192126434835493853f, 0.92387953251128673848f, 0.95694033573220893540f, 0.98078528040323043058f, 0.99518472667219681771f};

/* v * exp(-2 pi i Q / R), first quadrant (mul_w_q1) */
static inline cf mul_w_q1(int Q, int R, cf v) {
    const int idx = Q * (32 / R);
    if (idx == 0) return v;
    if (idx == 4) return cf_scale(add_mi(v, v), 0.70710678118654752440f);
    {
        const float c = kCos32[idx], sn = kSin32[idx];
        const float tx = v.x * c, ty = v.y * c;
        cf r = {fmaf(v.y, sn, tx), fmaf(v.x, -sn, ty)};
        return r;
    }
}

/* in-register DIF DFT of R points, natural order in and out; UR: the upper half of the inputs carries a pending factor -i (dft<R, UR>) */
static void dft(int R, int UR, cf* x) {
    if (R == 2) {
        const cf a = x[0], b = x[1];
        x[0] = UR ? add_mi(a, b) : cf_add(a, b);
        x[1] = UR ? sub_mi(a, b) : cf_sub(a, b);
        return;
    }
    cf a[16], b[16];
    for (int J = 0; J < R / 2; ++J) {
        const cf lo = x[J], hi = x[J + R / 2];
        a[J] = UR ? add_mi(lo, hi) : cf_add(lo, hi);
        const cf d = UR ? sub_mi(lo, hi) : cf_sub(lo, hi);
        b[J] = mul_w_q1(J % (R / 4), R, d);
    }
    dft(R / 2, 0, a);
    dft(R / 2, 1, b);
    for (int q = 0; q < R / 2; ++q) {
        x[2 * q] = a[q];
        x[2 * q + 1] = b[q];
    }
}

/* the first radix stage with the window multiply folded into its first butterfly layer (dft_win): x[J], J < R/2, are RAW samples with
 * their window values w[J]; x[J + R/2] are already windowed; a = fma(x_lo, w_lo, P_hi), d = fma(x_lo, w_lo, -P_hi) */
static void dft_win(int R, cf* x, const cf* w) {
    cf a[16], b[16];
    for (int J = 0; J < R / 2; ++J) {
        const cf hi = x[J + R / 2];
        a[J].x = fmaf(x[J].x, w[J].x, hi.x);
        a[J].y = fmaf(x[J].y, w[J].y, hi.y);
        cf d;
        d.x = fmaf(x[J].x, w[J].x, -hi.x);
        d.y = fmaf(x[J].y, w[J].y, -hi.y);
        b[J] = mul_w_q1(J % (R / 4), R, d);
    }
    dft(R / 2, 0, a);
    dft(R / 2, 1, b);
    for (int q = 0; q < R / 2; ++q) {
        x[2 * q] = a[q];
        x[2 * q + 1] = b[q];
    }
}

/* ---- split-radix register DFTs for R = 16 and 32 (jsg_stft_kernel.h: mul_w32, SrOdd, sr_rest, sr_dft, sr_dft_win); R <= 8: the radix-2 code ---- */
/* v * exp(-2 pi i E / 32): first-quadrant constants, the quadrant E / 8 in the signs and the component order; t = (+-c v_a, +-c v_b), r = fma(+-s v_c, .) */
static inline cf mul_w32(int E, cf v) {
    const int e = E % 8, q = E / 8;
    const float c = kCos32[e], sn = kSin32[e];
    cf r;
    if (q == 0) {
        const float tx = v.x * c, ty = v.y * c;
        r.x = fmaf(v.y, sn, tx);
        r.y = fmaf(-v.x, sn, ty);
    } else if (q == 1) {
        const float tx = v.y * c, ty = -v.x * c;
        r.x = fmaf(-v.x, sn, tx);
        r.y = fmaf(-v.y, sn, ty);
    } else if (q == 2) {
        const float tx = -v.x * c, ty = -v.y * c;
        r.x = fmaf(-v.y, sn, tx);
        r.y = fmaf(v.x, sn, ty);
    } else {
        const float tx = -v.y * c, ty = v.x * c;
        r.x = fmaf(v.x, sn, tx);
        r.y = fmaf(v.y, sn, ty);
    }
    return r;
}
static void sr_dft(int R, cf* x);
static void sr_rest(int R, cf* a, const cf* d, cf* x) {
    cf z1[8], z3[8];
    for (int n = 0; n < R / 4; ++n) {
        const cf p = d[n], q = d[n + R / 4];
        const cf u = add_mi(p, q), w = sub_mi(p, q);
        z1[n] = n == 0 ? u : mul_w32(n * (32 / R), u);
        z3[n] = n == 0 ? w : mul_w32((3 * n * (32 / R)) % 32, w);
    }
    sr_dft(R / 2, a);
    sr_dft(R / 4, z1);
    sr_dft(R / 4, z3);
    for (int k = 0; k < R / 2; ++k) x[2 * k] = a[k];
    for (int k = 0; k < R / 4; ++k) {
        x[4 * k + 1] = z1[k];
        x[4 * k + 3] = z3[k];
    }
}
static void sr_dft(int R, cf* x) {
    if (R <= 8) { dft(R, 0, x); return; }
    cf a[16], d[16];
    for (int J = 0; J < R / 2; ++J) {
        a[J] = cf_add(x[J], x[J + R / 2]);
        d[J] = cf_sub(x[J], x[J + R / 2]);
    }
    sr_rest(R, a, d, x);
}
static void sr_dft_win(int R, cf* x, const cf* w) {
    if (R <= 8) { dft_win(R, x, w); return; }
    cf a[16], d[16];
    for (int J = 0; J < R / 2; ++J) {
        const cf hi = x[J + R / 2];
        a[J].x = fmaf(x[J].x, w[J].x, hi.x);
        a[J].y = fmaf(x[J].y, w[J].y, hi.y);
        d[J].x = fmaf(x[J].x, w[J].x, -hi.x);
        d[J].y = fmaf(x[J].y, w[J].y, -hi.y);
    }
    sr_rest(R, a, d, x);
}

typedef struct { const char* name; int N, R1, R2, R3, L, twf; } mplan;
static const mplan kPlans[] = {   /* jsg_stft_kernel.h: Cfg512 .. Cfg8192 (radices, lanes per frame, factorised tables) */
    {"Cfg512", 512, 8, 8, 4, 32, 0},     {"Cfg1024", 1024, 8, 8, 8, 64, 0},    {"Cfg1024B", 1024, 16, 32, 1, 16, 0},
    {"Cfg2048", 2048, 16, 8, 8, 64, 1}, {"Cfg2048B", 2048, 32, 32, 1, 32, 0},
    {"Cfg4096", 4096, 16, 8, 16, 128, 0}, {"Cfg4096B", 4096, 8, 16, 16, 64, 1}, {"Cfg8192", 8192, 16, 16, 16, 256, 1}};

static const double two_pi = 6.283185307179586476925286766559;
static inline cf tw(long long num, long long den) {   /* (float cos, float sin) of -2 pi (num % den) / den, as fill_tables builds them */
    const double ang = -two_pi * (double)(num % den) / (double)den;
    cf r = {(float)cos(ang), (float)sin(ang)};
    return r;
}

/* one windowed frame -> |X[k]|^2, k = 0 .. N/2 (`win2`: the window table of the plan: float(double(w) * 0.5 * amp)) */
static void frame_power(const mplan* p, const float* x, const float* win2, float* pw, cf* E1, cf* E2, cf* Z) {
    const int N = p->N, M = N / 2, R1 = p->R1, R2 = p->R2, R3 = p->R3, L = p->L;
    const int two_stage = R3 == 1;
    cf t[32];
    /* stage 1: radix R1 over n1 (stride M / R1), twiddle W_{R1 R2}^{n2 k1} with n2 = t1 / R3 */
    for (int t1 = 0; t1 < M / R1; ++t1) {
        cf wl[16];
        for (int n1 = 0; n1 < R1; ++n1) {
            const int n = t1 + (M / R1) * n1;
            if (n1 < R1 / 2) {   /* lower inputs stay raw: their products are fused into the first butterfly layer */
                t[n1].x = x[2 * n];
                t[n1].y = x[2 * n + 1];
                wl[n1].x = win2[2 * n];
                wl[n1].y = win2[2 * n + 1];
            } else {
                t[n1].x = x[2 * n] * win2[2 * n];
                t[n1].y = x[2 * n + 1] * win2[2 * n + 1];
            }
        }
        sr_dft_win(R1, t, wl);
        E1[t1] = t[0];
        for (int k1 = 1; k1 < R1; ++k1) E1[k1 * (M / R1) + t1] = cmul(t[k1], tw((long long)(t1 / R3) * k1, (long long)R1 * R2));
    }
    /* stage 2: radix R2 over n2; two-stage plans end here (Z[k1 + R1 k2]), the others multiply by W_M^{n3 (k1 + R1 k2)} */
    for (int t2 = 0; t2 < M / R2; ++t2) {
        const int k1 = t2 / R3, n3 = t2 % R3;
        for (int n2 = 0; n2 < R2; ++n2) t[n2] = E1[k1 * (M / R1) + n2 * R3 + n3];
        sr_dft(R2, t);
        if (two_stage) {
            for (int k2 = 0; k2 < R2; ++k2) Z[k1 + R1 * k2] = t[k2];
            continue;
        }
        for (int k2 = 0; k2 < R2; ++k2) {
            cf w;
            if (p->twf) w = cmul(tw((long long)n3 * k2, M / R1), tw((long long)n3 * k1, M));   /* shared row B[n3][k2] times the lane's A */
            else w = tw((long long)n3 * (k1 + R1 * k2), M);
            E2[(k1 * R2 + k2) * R3 + n3] = cmul(t[k2], w);
        }
    }
    /* stage 3: radix R3 over n3 -> Z[t3 + R1 R2 k3] */
    if (!two_stage)
        for (int t3 = 0; t3 < R1 * R2; ++t3) {
            const int k1 = t3 % R1, k2 = t3 / R1;
            for (int n3 = 0; n3 < R3; ++n3) t[n3] = E2[(k1 * R2 + k2) * R3 + n3];
            sr_dft(R3, t);
            for (int k3 = 0; k3 < R3; ++k3) Z[t3 + R1 * R2 * k3] = t[k3];
        }
    Z[M] = Z[0];
    /* paired real-split post pass: the window carries 1/2; T = (-i W_N^k)(Z[k] - conj Z[M-k]), S = Z[k] + conj Z[M-k],
     * X[k] = S + T, X[M-k] = conj(S - T); both powers as re*re then fma(im, im, .) */
    for (int k = 0; k < M / 2; ++k) {
        const cf z = Z[k], q = Z[M - k];
        cf wp;
        if (p->twf) {
            const int ll = k % L, rho = k / L, Q = 64 / (N / L);
            const double angc = -two_pi * (double)ll / (double)N;
            const cf twC = {(float)sin(angc), (float)(-cos(angc))};
            const cf u = {kCos64[Q * rho], -kSin64[Q * rho]};
            wp = cmul(twC, u);
        } else {
            const double ang = -two_pi * (double)k / (double)N;
            wp.x = (float)sin(ang);
            wp.y = (float)(-cos(ang));
        }
        const cf S = add_conj(z, q);
        const cf T = cmul(sub_conj(z, q), wp);
        const float re0 = S.x + T.x, re1 = S.x - T.x, im0 = S.y + T.y, im1 = S.y - T.y;
        pw[k] = fmaf(im0, im0, re0 * re0);
        pw[M - k] = fmaf(im1, im1, re1 * re1);
    }
    {
        const cf z = Z[M / 2];
        pw[M / 2] = 4.0f * fmaf(z.x, z.x, z.y * z.y);
    }
}

/* ---- the pair plan (Cfg2048P): a channel PAIR as one complex sequence z = x1 + i x2 of N points; |X1[k]|^2 + |X2[k]|^2 = (|Z[k]|^2 +
 * |Z[N-k]|^2) / 2.  Decimation in time: E / O = the N/2-point transforms of the even / odd samples (two-stage 32 x 32 engine, the window --
 * one real value per complex sample -- folded into the first butterfly layer: PairFirstLayer), Z[k] = E[k] + W_N^k O[k], Z[k + N/2] = E[k] -
 * W_N^k O[k] with W_N^(k + N/4) = -i W_N^k taken from the lower half of the table.  Powers accumulated as (sum Re^2, sum Im^2). ---- */
static void pair_half_fft(const float* x1, const float* x2, const float* winp, int h, cf* E1, cf* Y) {
    const int M = 1024, R = 32;
    cf t[32], a[16], b[16];
    for (int t1 = 0; t1 < M / R; ++t1) {   /* stage 1: lane ll = t1 holds samples n = 2 (ll + 32 n1) + h */
        cf z[32];
        float w[32];
        for (int n1 = 0; n1 < R; ++n1) {
            const int n = 2 * (t1 + 32 * n1) + h;
            z[n1].x = x1[n];
            z[n1].y = x2[n];
            w[n1] = winp[n];
        }
        for (int J = 0; J < R / 2; ++J) {
            cf hi, d;
            hi.x = z[J + 16].x * w[J + 16];
            hi.y = z[J + 16].y * w[J + 16];
            a[J].x = fmaf(z[J].x, w[J], hi.x);
            a[J].y = fmaf(z[J].y, w[J], hi.y);
            d.x = fmaf(z[J].x, w[J], -hi.x);
            d.y = fmaf(z[J].y, w[J], -hi.y);
            b[J] = mul_w_q1(J % (R / 4), R, d);
        }
        dft(R / 2, 0, a);
        dft(R / 2, 1, b);
        for (int q = 0; q < R / 2; ++q) {
            t[2 * q] = a[q];
            t[2 * q + 1] = b[q];
        }
        E1[t1] = t[0];
        for (int k1 = 1; k1 < R; ++k1) E1[k1 * (M / R) + t1] = cmul(t[k1], tw((long long)t1 * k1, (long long)R * R));
    }
    for (int k1 = 0; k1 < M / R; ++k1) {   /* stage 2: no further twiddle, Y[k1 + 32 k2] */
        for (int n2 = 0; n2 < R; ++n2) t[n2] = E1[k1 * (M / R) + n2];
        dft(R, 0, t);   /* (the pair plan keeps the radix-2 transforms) */
        for (int k2 = 0; k2 < R; ++k2) Y[k1 + R * k2] = t[k2];
    }
}

/* one column of the pair plan: channels c0 .. c0 + nch - 1 (nch even) -> out[0 .. N/2]; `hs_scale` / `divisor` as the kernel's epilogue */
static void pair_column(const float* x, long long pitch, int c0, int nch, long long start, const float* winp, int exact_div, float scale, float divisor,
                        int exact_db, float* out, cf* E1, cf* Y0, cf* Y1, cf* accU, cf* accV) {
    const int N = 2048, M = 1024;
    for (int k = 0; k < M; ++k) accU[k].x = accU[k].y = accV[k].x = accV[k].y = 0.0f;
    for (int c = c0; c + 1 < c0 + nch; c += 2) {
        const float* x1 = x + (long long)c * pitch + start;
        const float* x2 = x1 + pitch;
        pair_half_fft(x1, x2, winp, 0, E1, Y0);
        pair_half_fft(x1, x2, winp, 1, E1, Y1);
        for (int k = 0; k < M; ++k) {
            const int upper = k >= M / 2;
            const cf O = cmul(Y1[k], tw(upper ? k - M / 2 : k, N));
            const cf E = Y0[k];
            const cf Z = upper ? add_mi(E, O) : cf_add(E, O);
            const cf Zm = upper ? sub_mi(E, O) : cf_sub(E, O);
            accU[k].x = fmaf(Z.x, Z.x, accU[k].x);
            accU[k].y = fmaf(Z.y, Z.y, accU[k].y);
            accV[k].x = fmaf(Zm.x, Zm.x, accV[k].x);
            accV[k].y = fmaf(Zm.y, Zm.y, accV[k].y);
        }
    }
    const float U0 = accU[0].x + accU[0].y, V0 = accV[0].x + accV[0].y;
    const float hs = 0.5f * scale;
    for (int k = 0; k <= M; ++k) {
        float o;
        if (k == M) o = V0 + V0;
        else {
            const float U = accU[k].x + accU[k].y;
            const float Vm = k == 0 ? U0 : accV[M - k].x + accV[M - k].y;
            o = U + Vm;
        }
        o = exact_div ? (o * 0.5f) / divisor : o * hs;
        out[k] = exact_db ? jsg_exact_db(o) : o;
    }
}

/* mix modes as include/jsg.h (JSG_MIX_*): 0 AbsMean, 1 Max, 2 Min, 3 Left, 4 Right, 101 Sum */
int jsg_mirror_columns(const char* plan_name, const float* x, long long pitch, int channels, int hop, int feedblocks, long long first_frame,
                       long long n_frames, const float* window, float power_scale, int mix_mode, int exact_db, float* out /* [n_frames][N/2+1] */) {
    if (!strcmp(plan_name, "Cfg2048P")) {   /* the pair plan: sum-type mixes over an even channel count */
        if (!x || !window || !out || channels < 2 || (channels & 1) || hop < 1 || feedblocks < 1 || (mix_mode != 0 && mix_mode != 101)) return -2;
        const int N = 2048, M = 1024, H = M + 1;
        float* winp = (float*)malloc(sizeof(float) * N);
        cf* buf = (cf*)malloc(sizeof(cf) * M * 5);
        const double amp = sqrt((double)power_scale);
        for (int n = 0; n < N; ++n) winp[n] = (float)((double)window[n] * amp);
        const int pow2 = (channels & (channels - 1)) == 0;
        const int exact_div = mix_mode == 0 && !pow2;
        const float scale = mix_mode == 0 ? 1.0f / (float)channels : 1.0f, divisor = mix_mode == 0 ? (float)channels : 1.0f;
        const int regular = (long long)hop * feedblocks == N;
        for (long long i = 0; i < n_frames; ++i) {
            const long long j = first_frame + i;
            const long long start = regular ? j * hop : (j / feedblocks) * N + (j % feedblocks) * hop;
            pair_column(x, pitch, 0, channels, start, winp, exact_div, scale, divisor, exact_db, out + i * H, buf, buf + M, buf + 2 * M, buf + 3 * M, buf + 4 * M);
        }
        free(winp); free(buf);
        return 0;
    }
    const mplan* p = NULL;
    for (unsigned i = 0; i < sizeof kPlans / sizeof kPlans[0]; ++i)
        if (!strcmp(kPlans[i].name, plan_name)) p = &kPlans[i];
    if (!p || !x || !window || !out || channels < 1 || hop < 1 || feedblocks < 1) return -2;
    const int N = p->N, M = N / 2, H = M + 1;
    float* win2 = (float*)malloc(sizeof(float) * N);
    float* pw = (float*)malloc(sizeof(float) * H);
    float* acc = (float*)malloc(sizeof(float) * H);
    cf* E1 = (cf*)malloc(sizeof(cf) * M);
    cf* E2 = (cf*)malloc(sizeof(cf) * M);
    cf* Z = (cf*)malloc(sizeof(cf) * (M + 1));
    const double a2 = 0.5 * sqrt((double)power_scale);
    for (int n = 0; n < N; ++n) win2[n] = (float)((double)window[n] * a2);
    int c0 = 0, c1 = channels;
    if (mix_mode == 3) c1 = 1;
    if (mix_mode == 4) { c0 = 1; c1 = 2; }
    const int regular = (long long)hop * feedblocks == N;
    for (long long i = 0; i < n_frames; ++i) {
        const long long j = first_frame + i;
        const long long start = regular ? j * hop : (j / feedblocks) * N + (j % feedblocks) * hop;
        const float init = mix_mode == 2 ? 1000000.0f : 0.0f;
        for (int k = 0; k < H; ++k) acc[k] = init;
        for (int c = c0; c < c1; ++c) {
            frame_power(p, x + (long long)c * pitch + start, win2, pw, E1, E2, Z);
            for (int k = 0; k < H; ++k) {
                if (mix_mode == 1) acc[k] = pw[k] > acc[k] ? pw[k] : acc[k];
                else if (mix_mode == 2) acc[k] = pw[k] < acc[k] ? pw[k] : acc[k];
                else if (c1 - c0 == 1) acc[k] = pw[k];
                else acc[k] = acc[k] + pw[k];
            }
        }
        if (mix_mode == 0 && channels > 1) {   /* m_powerfinal[kk] /= m_channels: exact scaling for powers of two, IEEE division otherwise */
            const int pow2 = (channels & (channels - 1)) == 0;
            const float scale = 1.0f / (float)channels, divisor = (float)channels;
            for (int k = 0; k < H; ++k) acc[k] = pow2 ? acc[k] * scale : acc[k] / divisor;
        }
        float* o = out + i * H;
        for (int k = 0; k < H; ++k) o[k] = exact_db ? jsg_exact_db(acc[k]) : acc[k];
    }
    free(win2); free(pw); free(acc); free(E1); free(E2); free(Z);
    return 0;
}

/* the shared logarithm on its own (tests: accuracy against the reference's double log10) */
void jsg_mirror_exact_db(const float* p, float* out, long long count) {
    for (long long i = 0; i < count; ++i) out[i] = jsg_exact_db(p[i]);
}
