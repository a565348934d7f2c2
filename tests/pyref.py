"""Pure-Python big-integer reference for BLS12-381 (test-side cross-check of the oracle; independent
of every C++/HIP implementation in the repo).  Affine arithmetic with modular inverses."""
P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001

G1 = (0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
      0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1)
G2 = ((0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8,
       0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e),
      (0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801,
       0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be))


class F1:
    """Fp as plain ints"""
    zero = 0
    one = 1
    @staticmethod
    def add(a, b): return (a + b) % P
    @staticmethod
    def sub(a, b): return (a - b) % P
    @staticmethod
    def mul(a, b): return a * b % P
    @staticmethod
    def inv(a): return pow(a, -1, P)
    @staticmethod
    def neg(a): return (-a) % P


class F2:
    """Fp2 as (c0, c1), u^2 = -1"""
    zero = (0, 0)
    one = (1, 0)
    @staticmethod
    def add(a, b): return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
    @staticmethod
    def sub(a, b): return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)
    @staticmethod
    def mul(a, b): return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
    @staticmethod
    def inv(a):
        n = pow(a[0] * a[0] + a[1] * a[1], -1, P)
        return (a[0] * n % P, (-a[1]) * n % P)
    @staticmethod
    def neg(a): return ((-a[0]) % P, (-a[1]) % P)


def ec_add(F, p, q):
    """affine add on y^2 = x^3 + b (a = 0); None = infinity"""
    if p is None: return q
    if q is None: return p
    if p[0] == q[0]:
        if p[1] == q[1] and p[1] != F.zero:
            x2 = F.mul(p[0], p[0])
            lam = F.mul(F.add(F.add(x2, x2), x2), F.inv(F.add(p[1], p[1])))
        else:
            return None
    else:
        lam = F.mul(F.sub(q[1], p[1]), F.inv(F.sub(q[0], p[0])))
    x3 = F.sub(F.sub(F.mul(lam, lam), p[0]), q[0])
    y3 = F.sub(F.mul(lam, F.sub(p[0], x3)), p[1])
    return (x3, y3)


def ec_mul(F, p, k):
    r = None
    while k:
        if k & 1: r = ec_add(F, r, p)
        p = ec_add(F, p, p)
        k >>= 1
    return r


def g1_unc(p):
    if p is None: return b"\x40" + bytes(95)
    return p[0].to_bytes(48, "big") + p[1].to_bytes(48, "big")


def g2_unc(p):
    if p is None: return b"\x40" + bytes(191)
    return p[0][1].to_bytes(48, "big") + p[0][0].to_bytes(48, "big") + p[1][1].to_bytes(48, "big") + p[1][0].to_bytes(48, "big")


def g1_comp(p):
    if p is None: return b"\xc0" + bytes(47)
    b = bytearray(p[0].to_bytes(48, "big"))
    b[0] |= 0x80
    if p[1] > (P - 1) // 2: b[0] |= 0x20
    return bytes(b)


def g2_comp(p):
    if p is None: return b"\xc0" + bytes(95)
    b = bytearray(p[0][1].to_bytes(48, "big") + p[0][0].to_bytes(48, "big"))
    b[0] |= 0x80
    y = p[1]
    big = (y[1] > (P - 1) // 2) if y[1] != 0 else (y[0] > (P - 1) // 2)
    if big: b[0] |= 0x20
    return bytes(b)


def fr_le(x): return (x % R).to_bytes(32, "little")
def fr_from_le(b): return int.from_bytes(b, "little")
