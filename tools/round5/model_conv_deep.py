"""Host model of conv_deep_kernel's ADDRESS arithmetic (csrc/conv_glds.hip, round 5): the DMA pieces' source / LDS
placement (swizzle, poison padding), the item stream, the fragment addresses, the pixel permutation and the un-permuting
epilogue, lane by lane in NumPy, against a direct 3x3 SAME convolution restricted to the workgroup's chunk range. Written
before the kernel saw a GPU (no GPU in the build container): a dev aid, not a test of the product."""
import numpy as np


def pix_in_block(l31, W):
    q, t = l31 >> 2, l31 & 3
    g = (0x96 >> q) & 1
    k = (q >> 1) * 4 + t
    if W == 16:
        return (k >> 3) * 16 + ((k >> 2) & 1) * 8 + (k & 3) + 4 * g
    return (k >> 2) * 8 + (k & 3) + 4 * g


def run(W, B, C0, C1, Cout, ks, seed=0):
    H = W; HW = W * W; BM, BN = 256, 128
    IPT = BM // HW; PW = 20 if W == 16 else 12; PH = H + 2; IMG = PH * PW
    NP = 3 if W == 16 else 4; PROWS = NP * 8 * 16; PBUF = PROWS * 64
    WITEM = BN * 64; WSTAGE = 2 * WITEM
    rng = np.random.RandomState(seed)
    M = B * HW
    assert M % 256 == 0 and Cout % 128 == 0 and C0 % 64 == 0 and C1 % 64 == 0
    Cin = C0 + C1
    x0 = rng.randint(-3, 4, (M, C0)).astype(np.float64)
    x1 = rng.randint(-3, 4, (M, C1)).astype(np.float64) if C1 else np.zeros((M, 0))
    wgt = rng.randint(-2, 3, (9, Cout, Cin)).astype(np.float64)          # packed forward layout [tap][Cout][Cin]
    nch0, nchunks = C0 // 64, Cin // 64
    tiles_m, tiles_n = M // BM, Cout // BN
    out = np.zeros((ks, M, Cout))
    for kz in range(ks):
        c_begin, c_end = kz * nchunks // ks, (kz + 1) * nchunks // ks
        for mt in range(tiles_m):
            for nt in range(tiles_n):
                m0, n0 = mt * BM, nt * BN
                acc = np.zeros((8, 64, 2, 2, 16))                        # wave, lane, i, j, r
                for c in range(c_begin, c_end):
                    s1 = c >= nch0
                    cb = ((c - nch0) if s1 else c) * 64
                    src = x1 if s1 else x0
                    wcol = (C0 if s1 else 0) + cb
                    # ---- the two half patches via the DMA pieces
                    patch = np.zeros((2, PROWS, 4, 8))                    # [half][row][physical slot][8 ch]
                    for half in range(2):
                        for wave in range(8):
                            for k in range(NP):
                                for lane in range(64):
                                    drow, dslot = lane >> 2, lane & 3
                                    pr = (wave + 8 * k) * 16 + drow
                                    img, rem = divmod(pr, IMG); py, px = divmod(rem, PW)
                                    v = img < IPT and 1 <= py <= H and 1 <= px <= W
                                    logical = dslot ^ ((pr >> 2) & 3)
                                    # LDS destination: piece (wave + 8k) * 1024 + lane * 16 -> row pr, physical slot dslot
                                    if v:
                                        pix = m0 + img * HW + (py - 1) * W + (px - 1)
                                        ch = cb + half * 32 + logical * 8
                                        patch[half, pr, dslot] = src[pix, ch:ch + 8]
                    for idx in range(18):                                 # items of the chunk
                        half, tap = (1, idx - 9) if idx >= 9 else (0, idx)
                        ky, kx = divmod(tap, 3)
                        # weights of the item via the DMA pieces: rows wave*16 + drow
                        witem = np.zeros((BN, 4, 8))
                        for wave in range(8):
                            for lane in range(64):
                                drow, dslot = lane >> 2, lane & 3
                                rl = wave * 16 + drow
                                logical = dslot ^ ((rl >> 2) & 3)
                                col = wcol + half * 32 + logical * 8
                                witem[rl, dslot] = wgt[tap, n0 + rl, col:col + 8]
                        for ksx in range(2):
                            for wave in range(8):
                                wn, wm = wave & 1, wave >> 1
                                # the MFMA itself: D[row][col] += sum_k A[row][k] B[k][col], k = 16 = (fh, 8)
                                A = np.zeros((2, 32, 16)); Bm = np.zeros((2, 16, 32))
                                for lane in range(64):
                                    fh, l31 = lane >> 5, lane & 31
                                    pib = pix_in_block(l31, W)
                                    for i in range(2):
                                        row = wn * 64 + i * 32 + l31
                                        byte = (row * 64 + ((fh ^ ((l31 >> 2) & 3)) << 4)) ^ (ksx << 5)
                                        A[i, l31, 8 * fh:8 * fh + 8] = witem[byte // 64, (byte % 64) // 16]
                                    for j in range(2):
                                        ml = wm * 64 + j * 32 + pib
                                        img, rem = divmod(ml, HW); oy, ox = divmod(rem, W)
                                        prow = img * IMG + (oy + ky) * PW + ox + kx
                                        byte = (prow * 64 + ((fh ^ ((prow >> 2) & 3)) << 4)) ^ (ksx << 5)
                                        Bm[j, 8 * fh:8 * fh + 8, l31] = patch[half, byte // 64, (byte % 64) // 16]
                                for i in range(2):
                                    for j in range(2):
                                        D = A[i] @ Bm[j]                 # [32 rows (channels)][32 cols (MFMA columns)]
                                        for lane in range(64):
                                            fh, l31 = lane >> 5, lane & 31
                                            for r in range(16):
                                                acc[wave, lane, i, j, r] += D[8 * (r // 4) + 4 * fh + (r % 4), l31]
                # ---- epilogue: staged rows (j*32 + pib), columns i*32 + 8q + 4fh + e
                for wave in range(8):
                    wn, wm = wave & 1, wave >> 1
                    stage = np.zeros((64, 64))
                    for lane in range(64):
                        fh, l31 = lane >> 5, lane & 31
                        pib = pix_in_block(l31, W)
                        for i in range(2):
                            for j in range(2):
                                for r in range(16):
                                    stage[j * 32 + pib, i * 32 + 8 * (r // 4) + 4 * fh + (r % 4)] = acc[wave, lane, i, j, r]
                    out[kz, m0 + wm * 64:m0 + wm * 64 + 64, n0 + wn * 64:n0 + wn * 64 + 64] = stage
    got = out.sum(0)
    # reference: 3x3 SAME convolution
    x = np.concatenate([x0, x1], 1).reshape(B, H, W, Cin)
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
    ref = np.zeros((B, H, W, Cout))
    for tap in range(9):
        ky, kx = divmod(tap, 3)
        ref += xp[:, ky:ky + H, kx:kx + W, :] @ wgt[tap].T
    err = np.abs(got - ref.reshape(M, Cout)).max()
    print("W=%d B=%d C0=%d C1=%d Cout=%d ks=%d: max |model - conv| = %g" % (W, B, C0, C1, Cout, ks, err))
    return err


if __name__ == "__main__":
    assert run(8, 4, 64, 0, 128, 1) == 0
    assert run(16, 1, 64, 64, 128, 2) == 0
    assert run(8, 8, 128, 64, 256, 3) == 0
