"""Wall time of zd_generate WITH a host callback (records cross PCIe and are handed to the consumer plane by plane):
    python scripts/deliver_time.py [ppd] [format]        (ZD_LIB_PATH selects the library build)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zeldovich_plt_amd.api as zd
WMAP = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "wmap1new.pow")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
fmt = sys.argv[2] if len(sys.argv) > 2 else "RVZel"
L = zd.load_library()
ps = zd.PowerSpectrum.from_file(WMAP, 720.0)
p = zd.make_params(n, icformat=fmt)
st = zd.ZdStats()
seen = [0, 0]
def _cb(user, z, nrec, recp, densp):
    # a consumer that touches the plane (sum of the first/last record words) without holding it
    seen[0] += 1
    seen[1] ^= C.cast(recp, C.POINTER(C.c_uint64))[0] ^ C.cast(recp, C.POINTER(C.c_uint64))[nrec * zd.RECORD_DTYPES[fmt].itemsize // 8 - 1]
    return 0
cb = zd.SLAB_CB(_cb)
for rep in range(2):
    seen[0] = 0
    t0 = time.time()
    rc = L.zd_generate(C.byref(p), C.byref(ps.pk), None, 0, cb, None, C.byref(st))
    dt = time.time() - t0
    print("PPD=%d %s  lib=%s  rc=%d planes=%d  wall %.2f s (library-reported %.2f s)  %.1f GB over PCIe -> %.1f GB/s" % (
        n, fmt, os.path.basename(zd.LIB_PATH), rc, seen[0], dt, st.seconds_total, n ** 3 * zd.RECORD_DTYPES[fmt].itemsize / 1e9,
        n ** 3 * zd.RECORD_DTYPES[fmt].itemsize / 1e9 / st.seconds_total), flush=True)
