"""ctypes binding of include/zeldovich_hip.h — the host-side mirror of the reference's call sites
(Parameters / PowerSpectrum / ZeldovichZ + ZeldovichXY, src/zeldovich.cpp:848-1032).

This module is plumbing only: every number is produced by libzeldovich_hip.so (HIP, gfx950).  There is
no CPU fallback — if the library is missing, or no GPU is present when a compute entry point is
called, it fails loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ZD_LIB_PATH: A/B-testing another build of the same library (tuning only)
LIB_PATH = os.environ.get("ZD_LIB_PATH") or os.path.join(_HERE, "csrc", "build", "libzeldovich_hip.so")

ICFORMATS = {"Zeldovich": 0, "RVZel": 1, "RVdoubleZel": 2, "ZelSimple": 3}
RECORD_DTYPES = {
    "Zeldovich": np.dtype([("ijk", "<u2", 3), ("pad", "<u2"), ("d", "<f8", 3)]),
    "RVZel": np.dtype([("ijk", "<u2", 3), ("pad", "<u2"), ("d", "<f4", 3), ("v", "<f4", 3)]),
    "RVdoubleZel": np.dtype([("ijk", "<u2", 3), ("pad", "<u2"), ("d", "<f8", 3), ("v", "<f8", 3)]),
    "ZelSimple": np.dtype([("d", "<f4", 3)]),
}
KERNEL_NAMES = ("k_gen", "k_zfft", "k_yfft", "k_xfft", "z_stage", "exchange_wait")


class ZdParams(C.Structure):
    _fields_ = [
        ("ppd", C.c_int64), ("numblock", C.c_int32), ("cpd", C.c_int32), ("boxsize", C.c_double),
        ("fundamental", C.c_double), ("nyquist", C.c_double), ("k_cutoff", C.c_double),
        ("f_cluster", C.c_double), ("z_initial", C.c_double), ("PLT_target_z", C.c_double),
        ("seed", C.c_int64), ("corner_modes", C.c_int32), ("qdensity", C.c_int32),
        ("qoneslab", C.c_int32), ("qonemode", C.c_int32), ("one_mode", C.c_int32 * 3),
        ("qPLT", C.c_int32), ("qPLTrescale", C.c_int32), ("icformat", C.c_int32),
        ("stream_factor", C.c_int32), ("profile", C.c_int32),
        ("store_mode", C.c_int32), ("serial_z", C.c_int32), ("ngpu", C.c_int32), ("exchange_planes", C.c_int32),
        ("f_NL", C.c_double), ("n_s", C.c_double), ("Omega_M", C.c_double),
        ("version", C.c_int32), ("pass_groups", C.c_int32),
    ]


class ZdPk(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("x", C.POINTER(C.c_double)), ("y", C.POINTER(C.c_double)),
        ("y2", C.POINTER(C.c_double)), ("normalization", C.c_double), ("Pk_smooth2", C.c_double),
        ("fixed_power", C.c_int32), ("is_powerlaw", C.c_int32), ("powerlaw_index", C.c_double),
        ("kmax", C.c_double), ("kmin", C.c_double),
    ]


class ZdStats(C.Structure):
    _fields_ = [
        ("max_disp", C.c_double * 3), ("density_variance", C.c_double), ("seconds_total", C.c_double),
        ("kernel_ms", C.c_double * 6), ("kernel_launches", C.c_int64 * 6),
        ("bytes_intermediate", C.c_int64), ("stream_factor", C.c_int32), ("modes_cached", C.c_int32),
        ("bytes_sent", C.c_int64), ("max_disp_index", C.c_int64 * 3),
    ]


class ZdParamStrings(C.Structure):
    _fields_ = [
        ("Pk_filename", C.c_char * 1024), ("output_dir", C.c_char * 1024),
        ("density_filename", C.c_char * 1024), ("PLT_filename", C.c_char * 1024),
        ("ICFormat", C.c_char * 64),
        ("Pk_scale", C.c_double), ("Pk_norm", C.c_double), ("Pk_sigma", C.c_double),
        ("Pk_sigma_ratio", C.c_double), ("Pk_smooth", C.c_double), ("Pk_powerlaw_index", C.c_double),
        ("qPk_fix_to_mean", C.c_int32), ("version", C.c_int32),
        ("f_NL", C.c_double), ("n_s", C.c_double), ("Omega_M", C.c_double), ("np", C.c_int64),
    ]


SLAB_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p)
GROUP_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p)
PASS_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p)

# every symbol include/zeldovich_hip.h declares (checked by tests/test_capi_symbols.py)
EXPORTED_SYMBOLS = [
    "zd_generate", "zd_choose_stream_factor", "zd_plan_create", "zd_plan_destroy", "zd_plan_narray", "zd_plan_store_mode",
    "zd_plan_stream_factor", "zd_plan_passes", "zd_plan_plane_step", "zd_plan_record_size", "zd_plan_exchange_bytes", "zd_plan_local_planes",
    "zd_plan_plane_z", "zd_plan_stage_z", "zd_plan_stage_y", "zd_plan_stage_x", "zd_plan_stats", "zd_comm_unique_id", "zd_comm_create", "zd_comm_destroy", "zd_plan_ring_bytes", "zd_plan_run_pass",
    "zd_params_from_file", "zd_pk_create_from_file", "zd_pk_create_powerlaw", "zd_pk_power",
    "zd_pk_sigmaR", "zd_pk_destroy", "zd_load_eigmodes", "zd_free", "zd_comm_abort", "zd_comm_traffic", "zd_choose_pass_groups", "zd_plan_run_passes", "zd_comm_probe", "zd_choose_pass_groups_measured",
    "zd_dispatch_report",
]
# test scaffolding: exists only in the -DZD_TESTING library (csrc/zd_testing.h, `make testing`), never in the product
TESTING_SYMBOLS = ["zd_test_draws", "zd_test_modes", "zd_test_modes_table", "zd_test_v1_words", "zd_test_generate_loopback", "zd_test_fail_rank",
                   "zd_test_fft", "zd_test_poison"]
STORE_MODES = {"auto": 0, "reference": 1, "packed": 2, "fields": 3}  # zd_params.store_mode (ZD_STORE_*)

_lib = None
_testing_lib = None
TESTING_LIB_PATH = os.environ.get("ZD_TESTING_LIB_PATH") or os.path.join(_HERE, "csrc", "build", "libzeldovich_hip_testing.so")


def load_library():
    """Load libzeldovich_hip.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        _lib = _load(LIB_PATH, testing=False)
    return _lib


def load_testing_library():
    """The -DZD_TESTING build of the same sources: the product plus the device test hooks (zd_test_*) and the in-process
    emulation of the RCCL calls.  Only tests/ use it."""
    global _testing_lib
    if _testing_lib is None:
        _testing_lib = _load(TESTING_LIB_PATH, testing=True)
    return _testing_lib


def _load(path, testing):
    if not os.path.exists(path):
        raise RuntimeError(
            "zeldovich_plt_amd: %s is missing — run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C zeldovich_plt_amd/csrc%s`; there is no CPU fallback." % (path, " testing" if testing else ""))
    try:
        # torch bundles its own HIP runtime; it must be the first one initialised in the process,
        # otherwise torch later reports "No HIP GPUs are available" (measured on the MI355X box)
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    L.zd_generate.argtypes = [C.POINTER(ZdParams), C.POINTER(ZdPk), vp, i64, SLAB_CB, vp, C.POINTER(ZdStats)]
    if testing:
        L.zd_test_generate_loopback.argtypes = L.zd_generate.argtypes
        L.zd_test_draws.argtypes = [i64, i64, vp, vp]
        L.zd_test_modes.argtypes = [C.POINTER(ZdParams), C.POINTER(ZdPk), i64, vp, vp]
        L.zd_test_modes_table.argtypes = [C.POINTER(ZdParams), C.POINTER(ZdPk), i64, vp, vp]
        L.zd_test_v1_words.argtypes = [i64, C.c_int32, vp]
        L.zd_test_fft.argtypes = [i32, i64, i32, vp, vp]
        L.zd_test_poison.argtypes = [C.c_int]
        L.zd_test_poison.restype = None
    L.zd_choose_stream_factor.argtypes = [C.POINTER(ZdParams), C.c_int, i64]
    L.zd_choose_pass_groups.argtypes = [C.POINTER(ZdParams), C.c_int, i64, C.POINTER(i32), C.POINTER(i32)]
    L.zd_plan_create.argtypes = [C.POINTER(ZdParams), C.POINTER(ZdPk), vp, i64, C.c_int, C.c_int, C.POINTER(vp)]
    L.zd_plan_destroy.argtypes = [vp]
    L.zd_plan_destroy.restype = None
    for name in ("zd_plan_narray", "zd_plan_store_mode", "zd_plan_stream_factor", "zd_plan_record_size", "zd_plan_passes", "zd_plan_plane_step"):
        getattr(L, name).argtypes = [vp]
        getattr(L, name).restype = i32
    for name in ("zd_plan_exchange_bytes", "zd_plan_local_planes"):
        getattr(L, name).argtypes = [vp]
        getattr(L, name).restype = i64
    L.zd_plan_plane_z.argtypes = [vp, C.c_int, i64]
    L.zd_plan_plane_z.restype = i64
    L.zd_plan_stage_z.argtypes = [vp, C.c_int, vp, vp]
    L.zd_plan_stage_y.argtypes = [vp, vp, vp]
    L.zd_plan_stage_x.argtypes = [vp, C.c_int, vp, i64, i64, vp, vp, vp]
    L.zd_plan_stats.argtypes = [vp, C.POINTER(ZdStats)]
    L.zd_comm_unique_id.argtypes = [vp]
    L.zd_comm_create.argtypes = [C.c_int, C.c_int, vp, C.POINTER(vp)]
    L.zd_comm_destroy.argtypes = [vp]
    L.zd_comm_destroy.restype = None
    L.zd_comm_abort.argtypes = [vp]
    L.zd_comm_abort.restype = None
    L.zd_comm_probe.argtypes = [vp, i64, i32, C.POINTER(C.c_double)]
    L.zd_choose_pass_groups_measured.argtypes = [C.POINTER(ZdParams), C.c_int, i64, C.c_double, C.POINTER(i32), C.POINTER(i32), C.POINTER(C.c_double)]
    L.zd_comm_traffic.argtypes = [vp, C.POINTER(i64), C.POINTER(i64), C.c_int]
    L.zd_comm_traffic.restype = None
    L.zd_plan_ring_bytes.argtypes = [vp, C.POINTER(i32)]
    L.zd_plan_ring_bytes.restype = i64
    L.zd_plan_run_pass.argtypes = [vp, vp, C.c_int, vp, vp, i64, GROUP_CB, vp, vp]
    L.zd_plan_run_passes.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, vp, i64, PASS_CB, vp, vp]
    L.zd_params_from_file.argtypes = [C.c_char_p, C.POINTER(ZdParams), C.POINTER(ZdParamStrings)]
    L.zd_pk_create_from_file.argtypes = [C.c_char_p, dbl, dbl, dbl, dbl, dbl, C.c_int, dbl, C.POINTER(vp), C.POINTER(ZdPk)]
    L.zd_pk_create_powerlaw.argtypes = [dbl, dbl, dbl, dbl, dbl, C.c_int, dbl, C.POINTER(vp), C.POINTER(ZdPk)]
    L.zd_pk_power.argtypes = [C.POINTER(ZdPk), dbl]
    L.zd_pk_power.restype = dbl
    L.zd_pk_sigmaR.argtypes = [C.POINTER(ZdPk), dbl]
    L.zd_pk_sigmaR.restype = dbl
    L.zd_pk_destroy.argtypes = [vp]
    L.zd_pk_destroy.restype = None
    L.zd_load_eigmodes.argtypes = [C.c_char_p, C.POINTER(vp), C.POINTER(i64)]
    L.zd_free.argtypes = [vp]
    L.zd_free.restype = None
    L.zd_dispatch_report.argtypes = [C.c_char_p, i64]
    L.zd_dispatch_report.restype = i64
    return L


def dispatch_report(testing=False):
    """{(launcher with its template arguments, source line): launches so far} of the product library (testing=True: of the
    -DZD_TESTING library, which counts its own launches) — zd_dispatch_report"""
    L = load_testing_library() if testing else load_library()
    n = L.zd_dispatch_report(None, 0)
    buf = C.create_string_buffer(int(n) + 4096)
    L.zd_dispatch_report(buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        cnt, ln, name = line.split("\t", 2)
        out[(name, int(ln))] = out.get((name, int(ln)), 0) + int(cnt)
    return out


def make_params(ppd, numblock=2, boxsize=720.0, seed=12346, k_cutoff=1.0, qPLT=0, qPLTrescale=0,
                PLT_target_z=0.0, z_initial=49.0, f_cluster=1.0, icformat="RVdoubleZel", qdensity=0,
                qoneslab=-1, qonemode=0, one_mode=(0, 0, 0), corner_modes=0, cpd=None, stream_factor=0,
                profile=0, f_NL=0.0, n_s=1.0, Omega_M=1.0, store_mode="auto", serial_z=0, ngpu=0, exchange_planes=0,
                version=2, pass_groups=0):
    """Parameters with the derived quantities of Parameters::setup (src/parameters.cpp:172-174); version = 1 (legacy
    mt19937 streams) adjusts NumBlock by k_cutoff as the reader does (src/parameters.cpp:129-141)."""
    p = ZdParams()
    p.ppd = ppd
    if version == 1 and k_cutoff != 1.0:
        numblock = int(numblock * k_cutoff + .5)
    p.version = version
    p.numblock = numblock
    p.cpd = cpd if cpd is not None else ppd
    p.boxsize = boxsize
    p.fundamental = 2.0 * np.pi / boxsize
    p.nyquist = np.pi / (boxsize / ppd)
    p.k_cutoff = k_cutoff
    p.f_cluster = f_cluster
    p.z_initial = z_initial
    p.PLT_target_z = PLT_target_z
    p.seed = int(seed)
    p.corner_modes = corner_modes
    p.qdensity = qdensity
    p.qoneslab = qoneslab
    p.qonemode = qonemode
    p.one_mode = (C.c_int32 * 3)(*one_mode)
    p.qPLT = qPLT
    p.qPLTrescale = qPLTrescale
    p.icformat = ICFORMATS[icformat]
    p.stream_factor = stream_factor
    p.profile = profile
    p.store_mode = STORE_MODES[store_mode] if isinstance(store_mode, str) else int(store_mode)
    p.serial_z = serial_z
    p.ngpu = ngpu
    p.exchange_planes = exchange_planes
    p.pass_groups = pass_groups
    p.f_NL, p.n_s, p.Omega_M = f_NL, n_s, Omega_M
    return p


def params_from_file(path):
    """Parameters(file): returns (ZdParams, ZdParamStrings); raises on invalid input."""
    L = load_library()
    p, s = ZdParams(), ZdParamStrings()
    if L.zd_params_from_file(os.fsencode(path), C.byref(p), C.byref(s)):
        raise ValueError("Invalid Parameters given: %s" % path)
    return p, s


class PowerSpectrum:
    """PowerSpectrum after InitFromFile / InitFromPowerLaw + Normalize (src/power_spectrum.cpp:130-223)."""

    def __init__(self, pk, handle):
        self.pk = pk
        self._h = handle

    @classmethod
    def from_file(cls, path, boxsize, Pk_scale=1.0, Pk_norm=8.0, Pk_sigma=0.0210839935761,
                  Pk_sigma_ratio=0.0, Pk_smooth=0.0, fix_to_mean=0):
        L = load_library()
        pk, h = ZdPk(), C.c_void_p()
        if L.zd_pk_create_from_file(os.fsencode(path), Pk_scale, Pk_norm, Pk_sigma, Pk_sigma_ratio, Pk_smooth,
                                    fix_to_mean, boxsize, C.byref(h), C.byref(pk)):
            raise RuntimeError("power spectrum file %r could not be loaded" % path)
        return cls(pk, h)

    @classmethod
    def from_powerlaw(cls, index, boxsize, Pk_norm=8.0, Pk_sigma=0.02, Pk_sigma_ratio=0.0, Pk_smooth=0.0,
                      fix_to_mean=0):
        L = load_library()
        pk, h = ZdPk(), C.c_void_p()
        if L.zd_pk_create_powerlaw(index, Pk_norm, Pk_sigma, Pk_sigma_ratio, Pk_smooth, fix_to_mean, boxsize,
                                   C.byref(h), C.byref(pk)):
            raise RuntimeError("power law spectrum could not be initialised")
        return cls(pk, h)

    def power(self, k):
        return load_library().zd_pk_power(C.byref(self.pk), float(k))

    def sigmaR(self, R):
        return load_library().zd_pk_sigmaR(C.byref(self.pk), float(R))

    def tables(self):
        n = self.pk.n
        return tuple(np.ctypeslib.as_array(a, (n,)).copy() for a in (self.pk.x, self.pk.y, self.pk.y2))

    def __del__(self):
        try:
            if self._h:
                load_library().zd_pk_destroy(self._h)
                self._h = None
        except Exception:
            pass


def _fmt_name(icformat):
    return [k for k, v in ICFORMATS.items() if v == icformat][0]


def _stats_dict(st):
    return dict(max_disp=np.array(list(st.max_disp)), density_variance=st.density_variance,
                seconds_total=st.seconds_total, kernel_ms=dict(zip(KERNEL_NAMES, list(st.kernel_ms))),
                kernel_launches=dict(zip(KERNEL_NAMES, list(st.kernel_launches))),
                bytes_intermediate=st.bytes_intermediate, stream_factor=st.stream_factor,
                modes_cached=bool(st.modes_cached), bytes_sent=st.bytes_sent,
                max_disp_index=np.array(list(st.max_disp_index)))


def generate(params, ps, eig=None, collect=True, loopback=False, testing=False):
    """ZeldovichZ + ZeldovichXY on cuda:0 through zd_generate.

    collect=True gathers every delivered plane (host callback, like WriteParticlesSlab) into
    records[z, y, x] (and density[z, y, x] when qdensity); collect=False uses the NULL sink.
    loopback=True (tests; params.ngpu >= 2): zd_test_generate_loopback — the RCCL branch of the exchange on an in-process
    emulation of its calls."""
    L = load_testing_library() if (loopback or testing) else load_library()  # testing: zd_generate of the -DZD_TESTING build (zd_test_poison)
    entry = L.zd_test_generate_loopback if loopback else L.zd_generate
    n = int(params.ppd)
    st = ZdStats()
    eigp, eig_ppd = (None, 0)
    if eig is not None:
        eig = np.ascontiguousarray(eig, dtype=np.float64)
        eigp, eig_ppd = eig.ctypes.data, eig.shape[0]
    out = {}
    if collect:
        dt = RECORD_DTYPES[_fmt_name(params.icformat)]
        rec = np.zeros((n, n * n), dtype=dt) if params.qdensity != 2 else None
        dens = np.zeros((n, n * n), dtype=np.float32) if params.qdensity else None
        seen = []

        def _cb(user, z, nrec, recp, densp):
            seen.append(int(z))
            if recp and rec is not None:
                C.memmove(rec[z].ctypes.data, recp, nrec * dt.itemsize)
            if densp and dens is not None:
                C.memmove(dens[z].ctypes.data, densp, nrec * 4)
            return 0

        cb = SLAB_CB(_cb)
        rc = entry(C.byref(params), C.byref(ps.pk), eigp, eig_ppd, cb, None, C.byref(st))
        out["records"] = None if rec is None else rec.reshape(n, n, n)
        out["density"] = None if dens is None else dens.reshape(n, n, n)
        out["planes_seen"] = seen
    else:
        rc = entry(C.byref(params), C.byref(ps.pk), eigp, eig_ppd, SLAB_CB(), None, C.byref(st))
    if rc:
        raise RuntimeError("zd_generate failed (rc=%d); see stderr" % rc)
    out.update(_stats_dict(st))
    return out


def generate_planes(params, ps, on_plane, eig=None):
    """zd_generate with a per-plane consumer: on_plane(z, records[y, x]) sees a VIEW valid only during the call (like the
    reference's single output buffer); nothing is accumulated here.  Returns the statistics + the number of planes."""
    L = load_library()
    n = int(params.ppd)
    dt = RECORD_DTYPES[_fmt_name(params.icformat)]
    st = ZdStats()
    eigp, eig_ppd = (None, 0)
    if eig is not None:
        eig = np.ascontiguousarray(eig, dtype=np.float64)
        eigp, eig_ppd = eig.ctypes.data, eig.shape[0]
    count = [0]

    def _cb(user, z, nrec, recp, densp):
        count[0] += 1
        if recp:
            buf = (C.c_char * (nrec * dt.itemsize)).from_address(recp)
            return int(on_plane(int(z), np.frombuffer(buf, dtype=dt).reshape(n, n)) or 0)  # non-zero aborts (WriteParticlesSlab failing)
        return 0

    cb = SLAB_CB(_cb)
    rc = L.zd_generate(C.byref(params), C.byref(ps.pk), eigp, eig_ppd, cb, None, C.byref(st))
    if rc:
        raise RuntimeError("zd_generate failed (rc=%d); see stderr" % rc)
    out = _stats_dict(st)
    out["planes"] = count[0]
    return out


class Plan:
    """Staged API: one rank's share of the Z and XY stages on device pointers (see the header)."""

    def __init__(self, params, ps, eig=None, rank=0, nranks=1, testing=False):
        self.L = load_testing_library() if testing else load_library()  # testing: the -DZD_TESTING build (zd_test_poison), tests only
        self.params = params
        self._eig = None if eig is None else np.ascontiguousarray(eig, dtype=np.float64)
        h = C.c_void_p()
        rc = self.L.zd_plan_create(C.byref(params), C.byref(ps.pk),
                                   None if self._eig is None else self._eig.ctypes.data,
                                   0 if self._eig is None else self._eig.shape[0], rank, nranks, C.byref(h))
        if rc:
            raise RuntimeError("zd_plan_create failed")
        self.h = h
        self.rank, self.nranks = rank, nranks
        self.narray = self.L.zd_plan_narray(h)
        self.store_mode = [k for k, v in STORE_MODES.items() if v == self.L.zd_plan_store_mode(h)][0]
        self.R = self.L.zd_plan_stream_factor(h)
        self.passes = self.L.zd_plan_passes(h)          # R, or R/2 when a pass carries two z-residues
        self.plane_step = self.L.zd_plan_plane_step(h)  # stage_x plane ranges are multiples of it
        self.record_size = self.L.zd_plan_record_size(h)
        self.exchange_bytes = self.L.zd_plan_exchange_bytes(h)
        self.local_planes = self.L.zd_plan_local_planes(h)

    def plane_z(self, residue, local_plane):
        return self.L.zd_plan_plane_z(self.h, residue, local_plane)

    def stage_z(self, residue, d_send, stream=0):
        if self.L.zd_plan_stage_z(self.h, residue, d_send, stream):
            raise RuntimeError("zd_plan_stage_z failed")

    def stage_y(self, d_recv, stream=0):
        if self.L.zd_plan_stage_y(self.h, d_recv, stream):
            raise RuntimeError("zd_plan_stage_y failed")

    def stage_x(self, residue, d_recv, plane0, nplanes, d_records, d_density=None, stream=0):
        if self.L.zd_plan_stage_x(self.h, residue, d_recv, plane0, nplanes, d_records, d_density, stream):
            raise RuntimeError("zd_plan_stage_x failed")

    def run_pass(self, residue, d_store, d_records, rec_planes, comm=None, consume=None, stream=0):
        """Z stage -> exchange (plane groups, overlapped) -> y/x stages inside the library (zd_plan_run_pass);
        consume(first_local_plane, nplanes, d_records_ptr, stream) is called per finished group of planes"""
        cb = GROUP_CB()
        if consume is not None:
            cb = GROUP_CB(lambda user, first, n, recp, st: int(consume(int(first), int(n), recp, st) or 0))
        if self.L.zd_plan_run_pass(self.h, comm.h if comm is not None else None, residue, d_store, d_records, rec_planes, cb, None, stream):
            raise RuntimeError("zd_plan_run_pass failed")

    def run_passes(self, first, step, d_store, d_store2, d_records, rec_planes, comm=None, consume=None, stream=0):
        """the passes first, first + step, ... in one call (zd_plan_run_passes); with a second store d_store2 and several ranks
        the passes are pipelined; consume(pass, first_local_plane, nplanes, d_records_ptr, stream)"""
        cb = PASS_CB()
        if consume is not None:
            cb = PASS_CB(lambda user, ps, first_, n, recp, st: int(consume(int(ps), int(first_), int(n), recp, st) or 0))
        if self.L.zd_plan_run_passes(self.h, comm.h if comm is not None else None, first, step, d_store, d_store2, d_records, rec_planes,
                                     cb, None, stream):
            raise RuntimeError("zd_plan_run_passes failed")

    def stats(self):
        st = ZdStats()
        if self.L.zd_plan_stats(self.h, C.byref(st)):
            raise RuntimeError("zd_plan_stats failed")
        return _stats_dict(st)

    def close(self):
        if self.h:
            self.L.zd_plan_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    """RCCL communicator of the library (one process per GPU): `exchange_id` makes rank 0's 128-byte id known to every
    rank, e.g. lambda b: torch.distributed broadcast of a uint8 tensor"""

    def __init__(self, rank, nranks, exchange_id):
        self.L = load_library()
        buf = (C.c_ubyte * 128)()
        if rank == 0 and self.L.zd_comm_unique_id(buf):
            raise RuntimeError("zd_comm_unique_id failed")
        raw = exchange_id(bytes(buf))
        idb = (C.c_ubyte * 128).from_buffer_copy(raw)
        h = C.c_void_p()
        if self.L.zd_comm_create(rank, nranks, idb, C.byref(h)):
            raise RuntimeError("zd_comm_create failed")
        self.h = h

    def traffic(self, reset=False):
        """(bytes sent to, bytes received from) other ranks"""
        a, b = C.c_int64(), C.c_int64()
        self.L.zd_comm_traffic(self.h, C.byref(a), C.byref(b), int(reset))
        return a.value, b.value

    def probe(self, bytes_per_peer=128 << 20, reps=3):
        """GB/s one link of this rank carries per direction while every peer is sent to and received from at once (0.0 without peers)"""
        r = C.c_double()
        if self.L.zd_comm_probe(self.h, int(bytes_per_peer), int(reps), C.byref(r)):
            raise RuntimeError("zd_comm_probe failed")
        return r.value

    def abort(self):
        if self.h:
            self.L.zd_comm_abort(self.h)

    def close(self):
        if self.h:
            self.L.zd_comm_destroy(self.h)
            self.h = None


# ---- device test hooks (the -DZD_TESTING library) ---------------------------------------------------------------------------
def test_draws(seed, kxyz):
    L = load_testing_library()
    k = np.ascontiguousarray(kxyz, dtype=np.int32).reshape(-1, 3)
    out = np.zeros((k.shape[0], 2), dtype=np.uint64)
    if L.zd_test_draws(int(seed), k.shape[0], k.ctypes.data, out.ctypes.data):
        raise RuntimeError("zd_test_draws failed")
    return out


def test_modes(params, ps, kxyz):
    L = load_testing_library()
    k = np.ascontiguousarray(kxyz, dtype=np.int32).reshape(-1, 3)
    out = np.zeros((k.shape[0], 2), dtype=np.float64)
    if L.zd_test_modes(C.byref(params), C.byref(ps.pk), k.shape[0], k.ctypes.data, out.ctypes.data):
        raise RuntimeError("zd_test_modes failed")
    return out[:, 0] + 1j * out[:, 1]


def test_modes_table(params, ps, kxyz):
    """D(k) and fundamental/k^2 through k_genf's table arithmetic; returns (complex D [n], float64 [n])"""
    L = load_testing_library()
    k = np.ascontiguousarray(kxyz, dtype=np.int32).reshape(-1, 3)
    out = np.zeros((k.shape[0], 3), dtype=np.float64)
    if L.zd_test_modes_table(C.byref(params), C.byref(ps.pk), k.shape[0], k.ctypes.data, out.ctypes.data):
        raise RuntimeError("zd_test_modes_table failed")
    return out[:, 0] + 1j * out[:, 1], out[:, 2]


def test_v1_words(seed, nblocks):
    """first 624 * nblocks words of the ZD_Version = 1 stream generator (gsl_rng_mt19937) for `seed`"""
    L = load_testing_library()
    out = np.zeros(624 * nblocks, dtype=np.uint32)
    if L.zd_test_v1_words(int(seed), int(nblocks), out.ctypes.data):
        raise RuntimeError("zd_test_v1_words failed")
    return out


def test_fft(x, axis_kind):
    """x: complex128 [lines, n]; returns the unnormalised inverse DFT of every line computed on the GPU."""
    L = load_testing_library()
    x = np.ascontiguousarray(x, dtype=np.complex128)
    lines, n = x.shape
    if axis_kind == 1:  # strided situation: device layout [n][lines]
        xin = np.ascontiguousarray(x.T)
        out = np.zeros_like(xin)
        rc = L.zd_test_fft(n, lines, 1, xin.ctypes.data, out.ctypes.data)
        out = np.ascontiguousarray(out.T)
    else:
        out = np.zeros_like(x)
        rc = L.zd_test_fft(n, lines, 0, x.ctypes.data, out.ctypes.data)
    if rc:
        raise RuntimeError("zd_test_fft failed")
    return out
