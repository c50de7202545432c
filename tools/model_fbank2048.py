"""Lane-level numpy model of the 2048-point real FFT data flow of fbank2048_kernel: lane / register / LDS
index maps (row pitches 68 and 17), twiddles, partner exchange, checked against numpy.fft, and the LDS bank
model of MI355X_MICROARCH.md applied to every access.  `python tools/model_fbank2048.py` prints the errors and
any bank conflict; tests/test_fbank2048_model.py runs it on CPU."""
import numpy as np
rng = np.random.default_rng(0)
x = rng.standard_normal(2048)
zc = x[0::2] + 1j * x[1::2]                      # z[n], n < 1024
lam = np.arange(64)
# registers z[lane][j] = z[lam + 64 j]
reg = np.stack([zc[lam + 64 * j] for j in range(16)], axis=1)      # [64][16]
W = lambda N, e: np.exp(-2j * np.pi * (e % N) / N)
# pass 1: FFT16 over j -> k1 ; twiddle W1024^(lam k1)
reg = np.fft.fft(reg, axis=1)
k1 = np.arange(16)
reg = reg * W(1024, lam[:, None] * k1[None, :])
# transpose A: write buf[k1*64 + (lam ^ 4(k1&7))]
buf = np.zeros(1088, complex)
def bank_check(name, addrs_dwords_per_lane, width, groups, nbanks):
    # addrs: [64] first dword address; width dwords per lane
    worst = 1
    for g in groups:
        cnt = {}
        for ln in g:
            for d in range(width):
                bnk = (addrs_dwords_per_lane[ln] + d) % nbanks
                cnt.setdefault(bnk, set()).add(addrs_dwords_per_lane[ln] + d)
        worst = max(worst, max(len(v) for v in cnt.values()))
    if worst > 1: print('  bank conflict x%d in %s' % (worst, name))
G32 = [list(range(0, 32)), list(range(32, 64))]
G16c = [list(range(i, i + 16)) for i in range(0, 64, 16)]
G128 = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
G128 = G128 + [[v + 32 for v in g] for g in G128]
for k in range(16):
    idx = k * 68 + lam
    buf[idx] = reg[:, k]
    bank_check('A write k1=%d' % k, 2 * idx, 2, G16c, 32)
# read for pass 2a: lane nu = (kq = nu>>2, bq = nu&3): z[4 i + a] = buf[kq*64 + ((16 a + bq + 4 i) ^ 4(kq&7))]
kq, bq = lam >> 2, lam & 3
r2 = np.zeros((64, 16), complex)
for i in range(4):
    for a in range(4):
        idx = kq * 68 + 16 * a + bq + 4 * i
        r2[:, 4 * i + a] = buf[idx]
        bank_check('A read i=%d a=%d' % (i, a), 2 * idx, 2, G32, 64)
# pass 2a: dft4 over a -> c ; twiddle W64^((bq + 4 i) c)
u = np.zeros((64, 4, 4), complex)   # [lane][i][c]
for i in range(4):
    u[:, i, :] = np.fft.fft(r2[:, 4 * i:4 * i + 4], axis=1)
    bb = bq + 4 * i
    u[:, i, :] *= W(64, bb[:, None] * np.arange(4)[None, :])
# transpose B: row r = 4 kq + c, pos = b ^ 2 s(r), s(r) = (r >> 1) & 7
buf2 = np.zeros(1088, complex)
for i in range(4):
    for c in range(4):
        idx = 68 * kq + 17 * c + bq + 4 * i
        buf2[idx] = u[:, i, c]
        bank_check('B write i=%d c=%d' % (i, c), 2 * idx, 2, G16c, 32)
# read: lane = row r, chunk t at float4 slot (t ^ s(r)) holds b = 2t, 2t+1
r3 = np.zeros((64, 16), complex)
for bb in range(16):
    idx = 17 * lam + bb
    r3[:, bb] = buf2[idx]
    bank_check('B read b=%d' % bb, 2 * idx, 2, G32, 64)
# pass 2b: FFT16 over b -> d: X[kq + 16 c + 64 d] with lane = 4 kq + c
X = np.fft.fft(r3, axis=1)
kappa = (lam >> 2) + 16 * (lam & 3)
full = np.zeros(1024, complex)
for d in range(16):
    full[kappa + 64 * d] = X[:, d]
ref = np.fft.fft(zc)
print('complex FFT max err', np.abs(full - ref).max())
# unpack: upper regs to bufx[(d - 8) * 64 + lane]; partner lane / row
bufx = np.zeros(1024, complex)
for d in range(8, 16):
    idx = (d - 8) * 64 + lam
    bufx[idx] = X[:, d]
kq, c = lam >> 2, lam & 3
nu_p = np.where(kq == 0, (4 - c) & 3, 4 * ((16 - kq) & 15) + (3 - c))
extra = np.where(lam == 0, 64, 0)
P = np.zeros(1025)
for d in range(8):
    idx = (7 - d) * 64 + nu_p + extra
    bank_check('X read d=%d' % d, 2 * idx, 2, G32, 64)
    zp = bufx[idx]
    zk = X[:, d]
    k = kappa + 64 * d
    w = W(2048, k)
    cc = zk + np.conj(zp); dd = -1j * (zk - np.conj(zp))
    a = 0.5 * (cc + dd * w); bv = 0.5 * np.conj(cc - dd * w)
    pk = np.abs(a) ** 2; pm = np.abs(bv) ** 2
    if d == 0:
        pk[0] = (X[0, 0].real + X[0, 0].imag) ** 2
        pm[0] = (X[0, 0].real - X[0, 0].imag) ** 2
    P[k] = pk
    P[1024 - k] = pm
    bank_check('P write pk d=%d' % d, k, 1, G32, 32)
    bank_check('P write pm d=%d' % d, 1024 - k, 1, G32, 32)
P[512] = np.abs(X[0, 8]) ** 2
refP = np.abs(np.fft.rfft(x)) ** 2
print('power max rel err', (np.abs(P - refP) / refP.max()).max())
