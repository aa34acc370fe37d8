"""Host model of conv_deepk_kernel's ADDRESS arithmetic (csrc/conv_deepk.hip, round 5): the DMA pieces' source / LDS placement
(swizzle, padding pixels), the item stream over the eight waves, the fragment addresses, the pixel permutation and the
un-permuting epilogue, lane by lane in NumPy, against a direct 3x3 SAME convolution -- plus the ds_read_b128 bank model of the
pixel fragments. Written before the kernel saw a GPU (there is none in the build container; both its first run and that of
its split-K predecessor conv_deep were parity-green): a dev aid, not a test of the product."""
import numpy as np


def pix_in_block(l31, W):
    q, t = l31 >> 2, l31 & 3
    g = (0x96 >> q) & 1
    k = (q >> 1) * 4 + t
    if W == 16:
        return (k >> 3) * 16 + ((k >> 2) & 1) * 8 + (k & 3) + 4 * g
    return (k >> 2) * 8 + (k & 3) + 4 * g


def run_k(W, B, C0, C1, Cout, seed=0):
    """conv_deepk_kernel (csrc/conv_deepk.hip): 128-pixel x 64-filter tiles, K split over the eight waves of the workgroup
    (unit = item slot up = wave / 2, k-step uk = wave % 2), intervals of four items, four half-patch buffers."""
    assert W == 16
    H = W; HW = W * W; BM, BN = 128, 64
    PW = 20; IMG = 10 * PW; PROWS = 256
    rng = np.random.RandomState(seed)
    M = B * HW
    Cin = C0 + C1
    assert M % 128 == 0 and Cout % 64 == 0 and C0 % 64 == 0 and C1 % 64 == 0 and (Cin // 64) % 2 == 0
    x0 = rng.randint(-3, 4, (M, C0)).astype(np.float64)
    x1 = rng.randint(-3, 4, (M, C1)).astype(np.float64) if C1 else np.zeros((M, 0))
    wgt = rng.randint(-2, 3, (9, Cout, Cin)).astype(np.float64)
    nch0, nchunks = C0 // 64, Cin // 64
    out = np.zeros((M, Cout))
    for mt in range(M // BM):
        for nt in range(Cout // BN):
            m0, n0 = mt * BM, nt * BN
            acc = np.zeros((8, 64, 2, 4, 16))
            for P in range(nchunks // 2):
                # the four half patches of the pair
                patch = np.zeros((4, PROWS, 4, 8))
                for h4 in range(4):
                    c = 2 * P + (h4 >> 1); s1 = c >= nch0
                    cb = ((c - nch0) if s1 else c) * 64
                    src = x1 if s1 else x0
                    for wave in range(8):
                        for k in range(2):
                            for lane in range(64):
                                drow, dslot = lane >> 2, lane & 3
                                pr = (wave + 8 * k) * 16 + drow
                                img, rem = divmod(pr, IMG); py, px = divmod(rem, PW)
                                iy = ((m0 >> 7) & 1) * 8 + py - 1
                                if img == 0 and 1 <= px <= 16 and 0 <= iy < 16:
                                    pix = (m0 >> 8) * 256 + iy * 16 + (px - 1)
                                    ch = cb + (h4 & 1) * 32 + (dslot ^ ((pr >> 2) & 3)) * 8
                                    patch[h4, pr, dslot] = src[pix, ch:ch + 8]
                for IV in range(9):
                    # weight stage of the interval: [item slot][64 rows][4 slots][8]
                    stage = np.zeros((4, 64, 4, 8))
                    for wave in range(8):
                        up, uk = wave >> 1, wave & 1
                        idx = 4 * IV + up; h4 = (idx * 57) >> 9; tap = idx - 9 * h4
                        assert h4 == idx // 9 and tap == idx % 9
                        c = 2 * P + (h4 >> 1); s1 = c >= nch0
                        wcol = (C0 if s1 else 0) + ((c - nch0) if s1 else c) * 64 + (h4 & 1) * 32
                        for g in range(2):
                            for lane in range(64):
                                drow, dslot = lane >> 2, lane & 3
                                rl = (uk * 2 + g) * 16 + drow
                                col = wcol + (dslot ^ ((rl >> 2) & 3)) * 8
                                stage[up, rl, dslot] = wgt[tap, n0 + rl, col:col + 8]
                    for wave in range(8):
                        up, uk = wave >> 1, wave & 1
                        idx = 4 * IV + up; h4 = (idx * 57) >> 9; tap = idx - 9 * h4
                        ky = (tap * 11) >> 5; kx = tap - 3 * ky
                        assert (ky, kx) == divmod(tap, 3)
                        A = np.zeros((2, 32, 16)); Bm = np.zeros((4, 16, 32))
                        for lane in range(64):
                            fh, l31 = lane >> 5, lane & 31
                            kslot = 2 * uk + fh
                            pib = pix_in_block(l31, W)
                            for i in range(2):
                                byte = l31 * 64 + ((kslot ^ ((l31 >> 2) & 3)) << 4) + i * 2048
                                A[i, l31, 8 * fh:8 * fh + 8] = stage[up, byte // 64, (byte % 64) // 16]
                            for j in range(4):
                                ml = j * 32 + pib
                                prow = (ml >> 4) * PW + (ml & 15) + ky * PW + kx
                                byte = prow * 64 + ((kslot ^ ((prow >> 2) & 3)) << 4)
                                Bm[j, 8 * fh:8 * fh + 8, l31] = patch[h4, byte // 64, (byte % 64) // 16]
                        for i in range(2):
                            for j in range(4):
                                D = A[i] @ Bm[j]
                                for lane in range(64):
                                    fh, l31 = lane >> 5, lane & 31
                                    for r in range(16):
                                        acc[wave, lane, i, j, r] += D[8 * (r // 4) + 4 * fh + (r % 4), l31]
            # reduction: wave w finishes block (i = w / 4, j = w % 4); epilogue mapping
            for wave in range(8):
                iw, jw = wave >> 2, wave & 3
                for lane in range(64):
                    fh, l31 = lane >> 5, lane & 31
                    m = m0 + jw * 32 + pix_in_block(l31, W)
                    for r in range(16):
                        c = n0 + iw * 32 + 8 * (r // 4) + 4 * fh + (r % 4)
                        out[m, c] = acc[:, lane, iw, jw, r].sum()
    x = np.concatenate([x0, x1], 1).reshape(B, H, W, Cin)
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
    ref = np.zeros((B, H, W, Cout))
    for tap in range(9):
        ky, kx = divmod(tap, 3)
        ref += xp[:, ky:ky + H, kx:kx + W, :] @ wgt[tap].T
    err = np.abs(out - ref.reshape(M, Cout)).max()
    print("deepk W=%d B=%d C0=%d C1=%d Cout=%d: max |model - conv| = %g" % (W, B, C0, C1, Cout, err))
    return err


def bank_check_k():
    """ds_read_b128 bank model of conv_deepk's pixel fragments (64-byte rows, four 16-lane groups, 256-byte bank row)."""
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    worst = 0
    for j in range(4):
        for tap in range(9):
            ky, kx = divmod(tap, 3)
            rows = []
            for l in range(32):
                ml = j * 32 + pix_in_block(l, 16)
                rows.append((ml >> 4) * 20 + (ml & 15) + ky * 20 + kx)
            for ks in range(4):
                for g in groups:
                    seen = {}
                    for l in g:
                        a = (rows[l] * 64 + ((ks ^ ((rows[l] >> 2) & 3)) << 4)) % 256
                        seen[a] = seen.get(a, 0) + 1
                    worst = max(worst, max(seen.values()))
    print("deepk pixel fragments: worst bank multiplicity", worst)
    return worst


if __name__ == "__main__":
    assert bank_check_k() == 1
    assert run_k(16, 1, 128, 0, 64) == 0
    assert run_k(16, 2, 64, 64, 128) == 0
