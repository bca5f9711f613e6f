"""CPU experiment behind k_fs_tables (TSVQ build, exact sequential f32 column sums on zero-mean columns).

Claim under test.  Let F be the map "incoming f32 sum -> sum after adding a segment's 64 addends one by one in f32".
For inputs written as S = c0 + 64 m + j in the integer grid of the input's binade (c0 a multiple of 64):
    F(c0 + 64 m + j) = F(c0 + j) + m * 64 ulp_in
for every m in [-Mlo, +Mhi], provided (1) the inputs c0 - 64 Mlo and c0 + 64 Mhi + 63 walk through the same sequence of
(sign, exponent) as c0 does -- by monotonicity of fl(s + x) in s everything in between then does too -- and (2) no
partial sum's binade is more than five above the input's (64 ulp_in is then an even multiple of every grid met).
The script checks the claim against brute force on N(0,1) columns and reports how often the true incoming sum (f32 chain)
falls inside the validity window built around the f64 prefix guess.
"""
import sys

import numpy as np

F = np.float32
MS = np.array([1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 24, 32, 40, 48, 64, 80, 96, 128, 160, 192, 256, 320, 384, 512, 640, 768,
               1024, 1280, 1536, 2048, 4096], np.int64)


def to_S(x):
    b = np.asarray(x, F).view(np.uint32).astype(np.int64)
    mag = (b & 0x7FFFFF) | 0x800000
    return np.where(b >> 31, -mag, mag), ((b >> 23) & 0xFF) - 127


def from_S(S, e):
    S = np.asarray(S, np.int64)
    mag = np.abs(S)
    ok = (mag >= (1 << 23)) & (mag < (1 << 24))
    bits = ((S < 0).astype(np.uint32) << 31) | (np.uint32(e + 127) << 23) | (mag & 0x7FFFFF).astype(np.uint32)
    return bits.astype(np.uint32).view(F), ok


def chain(starts, addends):
    """sequential f32 adds for a vector of starting values; returns finals and the (sign, exponent) itinerary keys"""
    s = np.asarray(starts, F).copy()
    keys = [s.view(np.uint32) >> 23]
    for a in addends:
        s = (s + F(a)).astype(F)
        keys.append(s.view(np.uint32) >> 23)
    return s, np.stack(keys)


def main(n=1 << 18, seed=1, nseg_check=4000):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(n).astype(F)
    true = np.add.accumulate(x, dtype=F)  # sequential f32
    exact = np.add.accumulate(x.astype(np.float64))
    nseg = n // 64
    stats = dict(crossing=0, hit=0, miss_window=0, miss_binade=0, period_bad=0, wrong=0, m_abs=[])
    checked = 0
    for t in range(1, nseg):
        lo, hi = t * 64, t * 64 + 64
        s_in_true = true[lo - 1]
        seg_keys = np.concatenate([[np.array(s_in_true, F).view(np.uint32) >> 23], true[lo:hi].view(np.uint32) >> 23])
        if np.all(seg_keys == seg_keys[0]):
            continue  # stays in its binade: the compact summary serves it
        stats["crossing"] += 1
        if checked >= nseg_check:
            continue
        checked += 1
        guess = F(exact[lo - 1])
        Sg, eg = to_S(guess)
        Sg, eg = int(Sg), int(eg)
        St, et = to_S(s_in_true)
        St, et = int(St), int(et)
        if et != eg:
            stats["miss_binade"] += 1
            continue
        c0 = Sg & ~63
        add = x[lo:hi]
        A, okA = from_S(c0 + np.arange(64), eg)
        assert okA.all()
        T, keysA = chain(A, add)
        R, keysR = chain(A[:1], add)
        emax = int(((keysR & 0xFF).max()) - 127)
        lo_c, ok_lo = from_S(c0 - 64 * MS, eg)
        hi_c, ok_hi = from_S(c0 + 64 * MS + 63, eg)
        _, k_lo = chain(lo_c, add)
        _, k_hi = chain(hi_c, add)
        same_lo = ok_lo & np.all(k_lo == keysR, axis=0)
        same_hi = ok_hi & np.all(k_hi == keysR, axis=0)
        mlo = -int(MS[np.nonzero(same_lo)[0].max()]) if same_lo.any() else 0
        mhi = int(MS[np.nonzero(same_hi)[0].max()]) if same_hi.any() else 0
        if emax - eg > 5:
            stats["period_bad"] += 1
            mlo = mhi = 0
        m, j = (St - c0) >> 6, St & 63
        stats["m_abs"].append(abs(m))
        if m < mlo or m > mhi:
            stats["miss_window"] += 1
            continue
        g = F(2.0) ** (eg - 17)
        pred = F(T[j] + F(m) * g)
        if pred.view(np.uint32) != true[hi - 1].view(np.uint32):
            stats["wrong"] += 1
            print("WRONG", t, m, j, mlo, mhi, pred, true[hi - 1])
        else:
            stats["hit"] += 1
        # brute force over the whole validity window for a few segments
        if checked % 200 == 0 and (mhi - mlo) <= 512:
            ms = np.arange(mlo, mhi + 1)
            allS = (c0 + 64 * ms[:, None] + np.arange(64)[None, :]).ravel()
            st, ok = from_S(allS, eg)
            assert ok.all()
            fin, _ = chain(st, add)
            want = (T[None, :] + (ms[:, None].astype(F) * g)).astype(F).ravel()
            bad = int((fin.view(np.uint32) != want.view(np.uint32)).sum())
            if bad:
                stats["wrong"] += bad
                print("BRUTE WRONG", t, bad, mlo, mhi)
    m_abs = np.array(stats.pop("m_abs"))
    print(f"n={n}: {nseg} segments, {stats['crossing']} cross a binade ({100.0 * stats['crossing'] / nseg:.1f} %); checked {checked}: {stats}")
    if m_abs.size:
        print("  |m| (periods of 64 ulps between guess and truth): median %d, 90 %% %d, max %d" %
              (np.median(m_abs), np.percentile(m_abs, 90), m_abs.max()))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
