#!/usr/bin/env python3
"""Design model of the wave-per-frame FFT used by jadespectrogram_amd/csrc/jsg_kernels.hip  (DEV TOOL).

One wavefront (64 lanes) transforms one real frame of N samples as an M = N/2 point complex FFT with
P = M/64 complex values per lane, in three register-resident stages of radix R1, R2, R3 (R1*R2*R3 = M)
separated by two wave-private LDS exchanges, followed by the real-split post pass.  R3 = 1 is the two-stage plan
(Cfg2048B): one exchange, read back as 16-byte pairs, and the second radix stage leaves bin ll + L*k2 in register k2.

This script (a) checks the index algebra numerically against numpy.fft, (b) counts LDS bank conflicts
of a layout under the gfx950 banking rules (MI355X_MICROARCH.md "LDS"), and (c) searches paddings.
The C++ plan builder (jsg_plan.cpp) implements exactly the tables defined here.
"""
import itertools
import sys

import numpy as np

LANES = 64


class Plan:
    HALF_OFFSET = 0
    def __init__(self, N, R1, R2, R3, S1=None, A=None, B=None, L=64, Z=1):
        global LANES
        LANES = L            # lanes that cooperate on one frame (64, or 32 = two frames per wave)
        self.L = L
        self.N, self.M = N, N // 2
        assert R1 * R2 * R3 == self.M
        self.R = (R1, R2, R3)
        self.P = self.M // L
        assert self.P % R1 == 0 and self.P % R2 == 0 and self.P % R3 == 0
        self.U = (self.P // R1, self.P // R2, self.P // R3)
        # exchange-1 layout: a1(k1, t1) = k1*S1 + t1
        self.S1 = S1 if S1 is not None else self.M // R1 + R3
        # exchange-2 layout: a2(k1,k2,n3) = k1*A + k2*B + n3
        self.B = B if B is not None else R3 + 1
        self.A = A if A is not None else R2 * self.B
        self.Z = Z
        self.two_stage = R3 == 1
        e2 = 0 if self.two_stage else (R1 - 1) * self.A + (R2 - 1) * self.B + (R3 - 1) * Z + 1
        self.lds_elems = max((R1 - 1) * self.S1 + self.M // R1, e2, self.M + 1)
        self.lds_elems += self.lds_elems & 1   # 16-byte multiple, as Cfg::LDS_ELEMS
        Plan.HALF_OFFSET = self.lds_elems

    # ---- addresses (in complex-element units) -------------------------------------------------
    def a1(self, k1, t1):
        return k1 * self.S1 + t1

    def a2(self, k1, k2, n3):
        return k1 * self.A + k2 * self.B + n3 * self.Z

    # ---- tables ------------------------------------------------------------------------------
    def tw1(self, u, k1, lane):
        R1, R2, R3 = self.R
        t1 = lane + LANES * u
        n2 = t1 // R3
        return np.exp(-2j * np.pi * (n2 * k1) / (R1 * R2))

    def tw2(self, v, k2, lane):
        R1, R2, R3 = self.R
        t2 = lane + LANES * v
        k1, n3 = t2 // R3, t2 % R3
        return np.exp(-2j * np.pi * (n3 * (k1 + R1 * k2)) / self.M)

    def tw2_factors(self, v, k2, lane):
        """Cfg::TWF: the stage-2 twiddle as (shared row B[n3][k2]) x (lane constant A[v]); must equal tw2()."""
        R1, R2, R3 = self.R
        t2 = lane + LANES * v
        k1, n3 = t2 // R3, t2 % R3
        return np.exp(-2j * np.pi * (n3 * k2) / (self.M // R1)), np.exp(-2j * np.pi * (n3 * k1) / self.M)

    def post(self, w, k3, lane):
        R1, R2, R3 = self.R
        k = lane + LANES * w + R1 * R2 * k3
        return -0.5j * np.exp(-2j * np.pi * k / self.N)

    # ---- numeric emulation (complex128) --------------------------------------------------------
    def run(self, frame):
        """frame: N real samples (already windowed).  Returns N/2+1 power values."""
        R1, R2, R3 = self.R
        U1, U2, U3 = self.U
        M, P = self.M, self.P
        z = frame[0::2] + 1j * frame[1::2]
        lds = np.zeros(self.lds_elems, dtype=np.complex128)
        reg = np.zeros((LANES, P), dtype=np.complex128)
        for lane in range(LANES):
            for m in range(P):
                reg[lane, m] = z[lane + LANES * m]
        # stage 1
        for lane in range(LANES):
            for u in range(U1):
                x = np.array([reg[lane, u + U1 * n1] for n1 in range(R1)])
                y = np.fft.fft(x)
                for k1 in range(R1):
                    lds[self.a1(k1, lane + LANES * u)] = y[k1] * self.tw1(u, k1, lane)
        # stage 2
        nxt = {} if self.two_stage else np.zeros_like(lds)   # two-stage plan: the results stay in registers (no layout)
        for lane in range(LANES):
            for v in range(U2):
                t2 = lane + LANES * v
                k1, n3 = t2 // R3, t2 % R3
                x = np.array([lds[self.a1(k1, n2 * R3 + n3)] for n2 in range(R2)])
                y = np.fft.fft(x)
                for k2 in range(R2):
                    nxt[(k1, k2) if self.two_stage else self.a2(k1, k2, n3)] = y[k2] * self.tw2(v, k2, lane)
        lds = nxt
        # stage 3
        Z = np.zeros(M + 1, dtype=np.complex128)
        for lane in range(LANES):
            for w in range(U3):
                t3 = lane + LANES * w
                k1, k2 = t3 % R1, t3 // R1
                x = np.array([lds[(k1, k2) if self.two_stage else self.a2(k1, k2, n3)] for n3 in range(R3)])
                y = np.fft.fft(x)
                for k3 in range(R3):
                    Z[t3 + R1 * R2 * k3] = y[k3]
        Z[M] = Z[0]
        # post pass
        out = np.zeros(M + 1)
        for lane in range(LANES):
            for w in range(U3):
                for k3 in range(R3):
                    k = lane + LANES * w + R1 * R2 * k3
                    zk, zp = Z[k], Z[M - k]
                    S = zk + np.conj(zp)
                    D = zk - np.conj(zp)
                    X = 0.5 * S + self.post(w, k3, lane) * D
                    out[k] = X.real ** 2 + X.imag ** 2
        out[M] = (Z[0].real - Z[0].imag) ** 2
        return out

    # ---- LDS bank model (8-byte elements) ------------------------------------------------------
    @staticmethod
    def _cycles(addrs, kind):
        """addrs: 64 element addresses (8-byte units).  ds_write_b64: 4 groups of 16 contiguous lanes, banks
        (a/4)%32 -> slot = elem % 16.  ds_read_b64: 2 groups of 32 lanes, banks (a/4)%64 -> slot = elem % 32."""
        if kind == "w":
            groups, mod = [range(g * 16, g * 16 + 16) for g in range(4)], 16
        else:
            groups, mod = [range(g * 32, g * 32 + 32) for g in range(2)], 32
        if len(addrs) == 32:   # two frames per wave: the second half-wave uses a disjoint LDS region
            addrs = list(addrs) + [a + Plan.HALF_OFFSET for a in addrs]
        cyc = 0
        ideal = 0
        for w0 in range(0, len(addrs), 64):   # frames spanning several wavefronts: one instruction per wavefront
            wa = addrs[w0:w0 + 64]
            for g in groups:
                slots = {}
                for l in g:
                    slots.setdefault(wa[l] % mod, set()).add(wa[l])
                cyc += max(len(s) for s in slots.values())
            ideal += len(groups)
        return cyc, ideal

    # ---- lane tables: 16-byte reads of the pair layout (jsg_kernels.hip, Cfg::tab_idx) -------------------------
    B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
                   [32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59], [36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63]]

    @staticmethod
    def _cycles_b128(addrs):
        """ds_read_b128: 4 groups of 16 lanes (MI355X_MICROARCH.md, LDS), banks (a/4) % 64 -> a 16-byte chunk c occupies
        banks 4c..4c+3: conflict-free when the 16 chunks of a group are distinct mod 16 (equal addresses broadcast)."""
        cyc = 0
        for g in Plan.B128_GROUPS:
            slots = {}
            for l in g:
                slots.setdefault((addrs[l] // 2) % 16, set()).add(addrs[l])
            cyc += max(len(s) for s in slots.values())
        return cyc, 4

    def table_conflicts(self):
        """Pair layout of the lane tables: element (j, e) of a [J][TL] table at ((j//2)*TL + e)*2 + j%2, read as 16 bytes per
        lane; stage-1 rows of R1 + 2 elements, k1 at column k1 - 1, one row per n2 = t1 // R3 (broadcast inside a row)."""
        R1, R2, R3 = self.R
        TL = max(self.L, 64)
        cyc = ideal = 0
        for w0 in range(0, TL, 64):
            c, i = self._cycles_b128([(0 * TL + (w0 + l)) * 2 for l in range(64)])            # window / stage-2 / post rows
            cyc += c; ideal += i
            for u in range(self.U[0]):
                rows = [(((w0 + l) % self.L + self.L * u) // R3) * (R1 + 2) for l in range(64)]   # pair (k1 = 1, 2) of every lane's row
                c, i = self._cycles_b128(rows)
                cyc += c; ideal += i
        return cyc, ideal

    def conflicts(self, verbose=False):
        R1, R2, R3 = self.R
        U1, U2, U3 = self.U
        tot = {}
        def acc(name, addrs, kind):
            c, ideal = self._cycles(addrs, kind)
            a = tot.setdefault(name, [0, 0])
            a[0] += c; a[1] += ideal
        for u in range(U1):
            for k1 in range(R1):
                acc("x1 write", [self.a1(k1, l + LANES * u) for l in range(LANES)], "w")
        for v in range(U2):
            if self.two_stage:   # the R2 values of a lane are neighbours in its row: 16-byte reads of pairs (n2, n2 + 1)
                for n2 in range(0, R2, 2):
                    addrs = [self.a1(l + LANES * v, n2) for l in range(LANES)]
                    addrs = addrs + [a + Plan.HALF_OFFSET for a in addrs] if len(addrs) == 32 else addrs
                    c, ideal = self._cycles_b128(addrs)
                    a = tot.setdefault("x1 read (16 B)", [0, 0])
                    a[0] += c; a[1] += ideal
                continue
            for n2 in range(R2):
                acc("x1 read", [self.a1((l + LANES * v) // R3, n2 * R3 + (l + LANES * v) % R3) for l in range(LANES)], "r")
            for k2 in range(R2):
                acc("x2 write", [self.a2((l + LANES * v) // R3, k2, (l + LANES * v) % R3) for l in range(LANES)], "w")
        for w in range(0 if self.two_stage else U3):
            for n3 in range(R3):
                acc("x2 read", [self.a2((l + LANES * w) % R1, (l + LANES * w) // R1, n3) for l in range(LANES)], "r")
        if verbose:
            for k, (c, i) in tot.items():
                print(f"   {k:9s}: {c:5d} LDS cycles (ideal {i})")
        return sum(c for c, _ in tot.values()), sum(i for _, i in tot.values())

    def check_injective(self):
        R1, R2, R3 = self.R
        s1 = {self.a1(k1, t1) for k1 in range(R1) for t1 in range(self.M // R1)}
        assert max(s1) < self.lds_elems
        if self.two_stage:
            return len(s1) == self.M
        s2 = {self.a2(k1, k2, n3) for k1 in range(R1) for k2 in range(R2) for n3 in range(R3)}
        assert max(s2) < self.lds_elems
        return len(s1) == self.M and len(s2) == self.M


def search(N, R1, R2, R3, L=64):
    M = N // 2
    best = None
    for padS in range(0, 33):
        S1 = M // R1 + padS
        for B in range(R3, R3 + 9):
            for padA in range(0, 33):
                A = R2 * B + padA
                p = Plan(N, R1, R2, R3, S1, A, B, L)
                if not p.check_injective():
                    continue
                c, i = p.conflicts()
                key = (c, p.lds_elems)
                if best is None or key < best[0]:
                    best = (key, (S1, A, B), i)
    return best


# N: (R1, R2, R3, lanes per frame, S1, AX, AY, AZ) -- the plans compiled into jsg_kernels.hip
CONFIGS = {512: (8, 8, 4, 32, 36, 4, 33, 1), 1024: (8, 8, 8, 64, 72, 9, 72, 2), 2048: (16, 8, 8, 64, 72, 65, 16, 2),
           "2048B": (32, 32, 1, 32, 34, 0, 0, 0),   # two-stage plan of the launches that mix >= 3 channels per column
           4096: (16, 8, 16, 128, 144, 1, 272, 17),
           "4096B": (8, 16, 16, 64, 272, 276, 17, 1),   # one wavefront per frame, factorised stage-2 / post tables (Cfg::TWF)
           8192: (16, 16, 16, 256, 272, 1, 272, 17)}

if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for N, cfg in CONFIGS.items():
        N = int(str(N).rstrip("B"))
        R1, R2, R3, L, S1, AX, AY, AZ = cfg
        if len(sys.argv) > 1 and sys.argv[1] == "search":
            print(N, (R1, R2, R3), "best (cycles, lds_elems), (S1,A,B), ideal:", search(N, R1, R2, R3, L))
            continue
        p = Plan(N, R1, R2, R3, S1=S1, A=AX, B=AY, L=L, Z=AZ)
        x = rng.standard_normal(N)
        ref = np.abs(np.fft.rfft(x)) ** 2
        got = p.run(x)
        err = np.max(np.abs(got - ref) / np.max(ref))
        print(f"N={N} radices={(R1, R2, R3)} L={L} P={p.P} S1={p.S1} AX={p.A} AY={p.B} AZ={p.Z} lds={p.lds_elems * 8} B "
              f"injective={p.check_injective()} max err={err:.2e}")
        p.conflicts(verbose=True)
        if L == 64 and p.P == 32:   # factorised tables of Cfg4096B: the product of the two factors is the full twiddle
            worst = max(abs(np.prod(p.tw2_factors(v, k2, lane)) - p.tw2(v, k2, lane))
                        for v in range(p.U[1]) for k2 in range(R2) for lane in range(L))
            print(f"   factorised stage-2 twiddles: max |B*A - W| = {worst:.1e}")
        tc, ti = p.table_conflicts()
        print(f"   tables   : {tc:5d} LDS cycles for the 16-byte reads of one table row and one stage-1 row per butterfly (ideal {ti})")
