import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, ROOT + "/tests", ROOT + "/oracle", ROOT + "/tests/golden"):
    sys.path.insert(0, p)
import crypto_amd
from oracle_c import OracleC
orc = OracleC()
n = 1001
braw = bytearray(orc.synth_bases(0x5EED0001, n))
braw[64 * 5:64 * 6] = bytes(64)
braw[64 * (n - 2):64 * (n - 1)] = braw[64 * 1:64 * 2]
braw = bytes(braw)
sraw = orc.synth_scalars(0x5EED0002, n)
want = orc.scalar_mul_each(braw, sraw)
for devs in (None, [0, 0]):
    e = crypto_amd.Engine(0) if devs is None else crypto_amd.Engine(devices=devs)
    h = e.upload(braw)
    r = e.scalar_mul_each(h, [int.from_bytes(sraw[32 * i:32 * i + 32], "big") for i in range(n)])
    got = b"".join(r.download())
    bad = [i for i in range(n) if got[64 * i:64 * i + 64] != want[64 * i:64 * i + 64]]
    print(devs, "mismatches:", len(bad), bad[:10], bad[-5:])
    e.close()
