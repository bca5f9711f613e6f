"""Offline fit of adder models to the probe data of tools/mfma_discover.py (numpy, exact integers via Python ints
where needed).  Model family: all addends aligned to a common exponent E, each converted to an integer on the grid
2^(E - W) with some rounding, summed exactly, the sum converted to f32 with some rounding."""
import sys
import numpy as np

def decode(z, fam):
    a, b, c, d = z[fam + '_a'], z[fam + '_b'], z[fam + '_c'], z[fam + '_d']
    ea = ((a >> 7) & 0xFF).astype(np.int64); eb = ((b >> 7) & 0xFF).astype(np.int64)
    ma = (a & 0x7F).astype(np.int64); mb = (b & 0x7F).astype(np.int64)
    sa = (a >> 15).astype(np.int64); sb = (b >> 15).astype(np.int64)
    cb = c.view(np.uint32).astype(np.int64)
    return dict(ea=ea, eb=eb, ma=ma, mb=mb, sp=sa ^ sb, ec=(cb >> 23) & 0xFF, mc=cb & 0x7FFFFF, sc=cb >> 31, d=d, c=c)

def round_shift(M, sh, mode):
    """integer M >= 0 (numpy int64 or object) shifted right by sh >= 0 bits with rounding `mode`; sh < 0: left shift"""
    M = M.astype(object); out = np.empty(M.shape, object)
    for idx in np.ndindex(M.shape):
        m = int(M[idx]); s = int(sh[idx])
        if s <= 0: out[idx] = m << (-s); continue
        if s > 80: q, r, half = 0, (1 if m else 0), 2
        else: q = m >> s; r = m & ((1 << s) - 1); half = 1 << (s - 1)
        if mode == 'trunc': pass
        elif mode == 'rne':
            if s <= 80 and (r > half or (r == half and (q & 1))): q += 1
        elif mode == 'rhu':
            if s <= 80 and r >= half: q += 1
        elif mode == 'sticky':  # truncate but OR a sticky bit into the LSB
            if r: q |= 1
        out[idx] = q
    return out

def to_f32(total, E, W, mode):
    """exact integer total on grid 2^(E-W) -> float32 bits with rounding mode"""
    out = np.empty(total.shape, np.float32)
    for idx in np.ndindex(total.shape):
        t = int(total[idx]); g = int(E[idx]) - W  # value = t * 2^g  (E unbiased)
        if t == 0: out[idx] = 0.0; continue
        s = t < 0; m = -t if s else t
        nb = m.bit_length()
        sh = nb - 24
        if sh > 0:
            q = m >> sh; r = m & ((1 << sh) - 1); half = 1 << (sh - 1)
            if mode == 'rne' and (r > half or (r == half and (q & 1))): q += 1
            if mode == 'rhu' and r >= half: q += 1
            if q == (1 << 24): q >>= 1; sh += 1
        else:
            q = m << (-sh)
        v = float(q) * 2.0 ** (g + sh)
        out[idx] = np.float32(-v if s else v)
    return out

def model(D, W, add_mode, fin_mode, prodE='raw', Wc=None, n=None):
    """returns predicted f32 for the first n trials"""
    sl = slice(0, n)
    ea, eb, ma, mb, sp, ec, mc, sc = (D[k][sl] for k in ('ea', 'eb', 'ma', 'mb', 'sp', 'ec', 'mc', 'sc'))
    live = (ea != 0) & (eb != 0)           # bf16 zero / subnormal operands -> no contribution (first guess)
    Mp = (128 + ma) * (128 + mb)           # 16-bit product significand, value Mp * 2^(ep - 14)
    ep = ea + eb - 254
    if prodE == 'norm': epn = ep + (Mp >= (1 << 15))
    else: epn = ep
    clive = ec != 0
    Mc = (1 << 23) + mc; ecu = ec - 127
    E = np.maximum(np.where(live, epn, -10**6).max(axis=1), np.where(clive, ecu, -10**6))
    # products: value = Mp * 2^(ep-14); on grid 2^(E-W): Mp * 2^(ep - 14 - E + W) -> shift right by (14 + E - W - ep)
    shp = 14 + E[:, None] - W - ep
    n_p = round_shift(np.where(live, Mp, 0), shp, add_mode)
    Wc_ = W if Wc is None else Wc
    shc = 23 + E - Wc_ - ecu
    n_c = round_shift(np.where(clive, Mc, 0), shc, add_mode)
    if Wc_ != W: n_c = np.array([int(x) << (W - Wc_) if W >= Wc_ else int(x) >> (Wc_ - W) for x in n_c], object)
    sgn_p = np.where(sp == 1, -1, 1).astype(object)
    tot = (n_p * sgn_p).sum(axis=1) + n_c * np.where(sc == 1, -1, 1).astype(object)
    return to_f32(tot, E, W, fin_mode)

if __name__ == '__main__':
    z = np.load('gpurun_out/mfma_probe.npz')
    fams = sys.argv[1].split(',') if len(sys.argv) > 1 else ['pair40']
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
    for fam in fams:
        D = decode(z, fam)
        d = D['d'][:n]
        for W in (21, 22, 23, 24, 25, 26):
            for add_mode in ('trunc', 'rne', 'rhu'):
                for fin_mode in ('trunc', 'rne'):
                    for prodE in ('raw', 'norm'):
                        p = model(D, W, add_mode, fin_mode, prodE, n=n)
                        ok = (p.view(np.uint32) == d.view(np.uint32)) | ((p == 0) & (d == 0))
                        if ok.mean() > 0.9:
                            print(f"{fam:10s} W={W} add={add_mode:5s} fin={fin_mode:5s} E={prodE:4s}: match {ok.mean():.4f}")
