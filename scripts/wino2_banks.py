#!/usr/bin/env python3
"""CPU model of conv_wino2.hip's data movement (no GPU): (1) every ds_read_b128 of the K loop, per 16-lane service group, counted for
LDS bank conflicts in both tile-group configurations; (2) the whole kernel emulated in numpy on a small layer — halo planes with the
swizzled columns, lane -> tile maps, B^T d B / G g G^T / A^T M A, the packed weight order — against a direct convolution.
Mirrors row_key / lane_tile / the loader index arithmetic of the kernel one to one."""
import numpy as np

GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
YA, YB, SA, SB = [0, 1, 2, 1], [2, 2, 1, 3], [1, 1, 1, 1], [-1, 1, -1, -1]
AT = [[1, 1, 1, 0], [0, 1, -1, -1]]
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])


def row_key(TGC, pl, pr, pc):
    return ((pc >> 1) + 4 * (pr + pl)) & 7 if TGC == 7 else ((pc >> 1) + 2 * pr) & 7


def lane_tile(TGC, t):
    if TGC == 7:
        tr, tc = t >> 3, t & 7
        return (tr, 6, False) if tc == 7 else (tr, tc, True)
    if t < 4: return 0, t, True
    if t < 12: return 1, t - 4, True
    if t < 16: return 0, t - 8, True
    if t < 20: return 3, t - 16, True
    if t < 28: return 2, t - 20, True
    return 3, t - 24, True


def lds_row(TGC, tgi, pl, pr, pc):
    PW = TGC + 1
    return tgi * 4 * 5 * PW + (pl * 5 + pr) * PW + pc


def bank_report(TGC):
    worst, tot, n = 0, 0, 0
    for dy in range(4):
        for dx in range(4):
            for g in range(4):
                for h in range(2):
                    for grp in GROUPS:
                        slots = {}
                        for t in grp:
                            tr, tc, _ = lane_tile(TGC, t)
                            pl, pr, pc = (dy & 1) * 2 + (dx & 1), tr + (dy >> 1), tc + (dx >> 1)
                            R = lds_row(TGC, 0, pl, pr, pc)
                            unit = R * 8 + ((2 * g + h) ^ row_key(TGC, pl, pr, pc))
                            slots.setdefault(unit % 16, set()).add(unit)
                        w = max(len(v) for v in slots.values())
                        worst, tot, n = max(worst, w), tot + w, n + 1
    return worst, tot / n


def emulate(TGC, B, H, W, Cin, Cout, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((Cout, 9, Cin)) / np.sqrt(9 * Cin)).astype(np.float32)
    # direct reference
    xp = np.zeros((B, H + 2, W + 2, Cin)); xp[:, 1:-1, 1:-1] = x
    ref = np.zeros((B, H, W, Cout))
    for ky in range(3):
        for kx in range(3):
            ref += np.einsum("bhwc,oc->bhwo", xp[:, ky:ky + H, kx:kx + W], w[:, ky * 3 + kx].astype(np.float64))
    # packed weights, as wino2_pack_weights
    NC = Cin // 32
    U = np.zeros((Cout // 64, NC, 16, 2, 4, 2, 32, 4))
    for co in range(Cout):
        for ci in range(Cin):
            u = G @ w[co, :, ci].reshape(3, 3).astype(np.float64) @ G.T
            U[co // 64, ci // 32, :, (co % 64) // 32, (ci % 32) // 8, (ci % 8) // 4, co % 32, ci % 4] = u.reshape(16)
    PW, RPT = TGC + 1, 4 * 5 * (TGC + 1)
    ht, wt = (H + 1) // 2, (W + 1) // 2
    tgy, tgx = (ht + 3) // 4, (wt + TGC - 1) // TGC
    n_tg = B * tgy * tgx
    out = np.full((B, H, W, Cout), np.nan)
    for pair in range((n_tg + 1) // 2):
        for tile_n in range(Cout // 64):
            Y = np.zeros((2, 2, 32, 2, 2, 64))                     # [tgi][mb][tile lane][a][b][channel in tile] (64: both halves for simplicity)
            for c in range(NC):
                halo = np.zeros((2 * RPT + 64, 8, 4))
                for R in range(2 * RPT):
                    g, rr = divmod(R, RPT); pl, r2 = divmod(rr, 5 * PW); pr, pc = divmod(r2, PW)
                    tg = 2 * pair + g
                    if tg >= n_tg: continue
                    n, rem = divmod(tg, tgx * tgy); gy, gx = divmod(rem, tgx)
                    y, xx = 8 * gy - 1 + 2 * pr + (pl >> 1), 2 * TGC * gx - 1 + 2 * pc + (pl & 1)
                    if 0 <= y < H and 0 <= xx < W:
                        for col in range(8):
                            lc = col ^ row_key(TGC, pl, pr, pc)
                            halo[R, col] = x[n, y, xx, 32 * c + 4 * lc: 32 * c + 4 * lc + 4]
                for f in range(16):
                    fi, fj = f >> 2, f & 3
                    for tgi in range(2):
                        for t in range(32):
                            tr, tc, live = lane_tile(TGC, t)
                            V = np.zeros(32)
                            for gq in range(4):
                                for h in range(2):
                                    def rd(dy, dx):
                                        pl, pr, pc = (dy & 1) * 2 + (dx & 1), tr + (dy >> 1), tc + (dx >> 1)
                                        R = lds_row(TGC, tgi, pl, pr, pc)
                                        return halo[R, (2 * gq + h) ^ row_key(TGC, pl, pr, pc)]
                                    v = SA[fi] * SA[fj] * rd(YA[fi], YA[fj]) + SA[fi] * SB[fj] * rd(YA[fi], YB[fj]) + \
                                        SB[fi] * SA[fj] * rd(YB[fi], YA[fj]) + SB[fi] * SB[fj] * rd(YB[fi], YB[fj])
                                    V[8 * gq + 4 * h: 8 * gq + 4 * h + 4] = v
                            Wm = np.zeros((64, 32))
                            for mb in range(2):
                                for gq in range(4):
                                    for kh in range(2):
                                        Wm[32 * mb: 32 * mb + 32, 8 * gq + 4 * kh: 8 * gq + 4 * kh + 4] = U[tile_n, c, f, mb, gq, kh]
                            M = Wm @ V
                            for a in range(2):
                                for b in range(2):
                                    Y[tgi, 0, t, a, b] += AT[a][fi] * AT[b][fj] * M
            for tgi in range(2):
                tg = 2 * pair + tgi
                if tg >= n_tg: continue
                n, rem = divmod(tg, tgx * tgy); gy, gx = divmod(rem, tgx)
                for t in range(32):
                    tr, tc, live = lane_tile(TGC, t)
                    if not live: continue
                    for a in range(2):
                        for b in range(2):
                            oy, ox = 2 * (4 * gy + tr) + a, 2 * (TGC * gx + tc) + b
                            if oy < H and ox < W:
                                assert np.isnan(out[n, oy, ox, 64 * tile_n]), "pixel written twice"
                                out[n, oy, ox, 64 * tile_n: 64 * tile_n + 64] = Y[tgi, 0, t, a, b]
    assert not np.isnan(out).any(), "pixel never written"
    return np.abs(out - ref).max()


if __name__ == "__main__":
    for TGC in (7, 8):
        print(f"TGC={TGC}: worst / mean LDS cycles per 16-lane group (1 = conflict-free):", bank_report(TGC))
    print("emulation TGC=7, 1x14x14x32->64  max err", emulate(7, 1, 14, 14, 32, 64))
    print("emulation TGC=8, 2x9x17x64->64   max err", emulate(8, 2, 9, 17, 64, 64))
    print("emulation TGC=7, 1x6x28x32->128  max err", emulate(7, 1, 6, 28, 32, 128))
