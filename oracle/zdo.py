"""TEST INFRASTRUCTURE ONLY: ctypes loader for oracle/_build/libzd_oracle.so (the plain-C CPU
restatement of the reference path) and, when present, oracle/_ref/libzd_ref.so (reference headers
compiled as they lie).  Importable only from tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke(); the product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libzd_oracle.so")
_REF = os.path.join(_HERE, "_ref", "libzd_ref.so")

ICFORMATS = {"Zeldovich": 0, "RVZel": 1, "RVdoubleZel": 2, "ZelSimple": 3}
RECORD_DTYPES = {
    "Zeldovich": np.dtype([("ijk", "<u2", 3), ("pad", "<u2"), ("d", "<f8", 3)]),
    "RVZel": np.dtype([("ijk", "<u2", 3), ("pad", "<u2"), ("d", "<f4", 3), ("v", "<f4", 3)]),
    "RVdoubleZel": np.dtype([("ijk", "<u2", 3), ("pad", "<u2"), ("d", "<f8", 3), ("v", "<f8", 3)]),
    "ZelSimple": np.dtype([("d", "<f4", 3)]),
}


def build(force=False):
    if force or not os.path.exists(_LIB):
        subprocess.check_call(["make", "-C", _HERE, "_build/libzd_oracle.so"])
    return _LIB


class Pcg(C.Structure):
    _fields_ = [("hi", C.c_uint64), ("lo", C.c_uint64)]


class Params(C.Structure):
    _fields_ = [
        ("ppd", C.c_int64), ("numblock", C.c_int), ("cpd", C.c_int), ("boxsize", C.c_double),
        ("separation", C.c_double), ("fundamental", C.c_double), ("nyquist", C.c_double),
        ("k_cutoff", C.c_double), ("qdensity", C.c_int), ("qoneslab", C.c_int), ("seed", C.c_int),
        ("f_cluster", C.c_double), ("qonemode", C.c_int), ("one_mode", C.c_int * 3),
        ("qPLT", C.c_int), ("qPLTrescale", C.c_int), ("PLT_target_z", C.c_double),
        ("z_initial", C.c_double), ("CornerModes", C.c_int), ("icformat", C.c_int),
        ("nthreads", C.c_int), ("f_NL", C.c_double), ("n_s", C.c_double), ("Omega_M", C.c_double), ("version", C.c_int),
    ]


class Pk(C.Structure):
    _fields_ = [
        ("n", C.c_int), ("x", C.POINTER(C.c_double)), ("y", C.POINTER(C.c_double)),
        ("y2", C.POINTER(C.c_double)), ("normalization", C.c_double), ("Pk_smooth2", C.c_double),
        ("fixed_power", C.c_int), ("is_powerlaw", C.c_int), ("powerlaw_index", C.c_double),
        ("kmin", C.c_double), ("kmax", C.c_double), ("Rnorm", C.c_double),
        ("primordial_norm", C.c_double), ("n_s", C.c_double),
    ]


class Stats(C.Structure):
    _fields_ = [("max_disp", C.c_double * 3), ("density_variance", C.c_double),
                ("t_stage1", C.c_double), ("t_store", C.c_double), ("t_load", C.c_double),
                ("t_fft2d", C.c_double), ("t_write", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.zdo_pcg_next.restype = C.c_uint64
        L.zdo_pcg_distance.restype = C.c_uint64
        L.zdo_u01.restype = C.c_double
        L.zdo_u01.argtypes = [C.c_uint64]
        L.zdo_spline_val.restype = C.c_double
        L.zdo_power.restype = C.c_double
        L.zdo_power.argtypes = [C.POINTER(Pk), C.c_double]
        L.zdo_sigmaR.restype = C.c_double
        L.zdo_sigmaR.argtypes = [C.POINTER(Pk), C.c_double]
        L.zdo_pk_from_file.argtypes = [C.POINTER(Pk), C.c_char_p] + [C.c_double] * 5 + [C.c_int, C.c_double]
        L.zdo_pk_from_powerlaw.argtypes = [C.POINTER(Pk)] + [C.c_double] * 5 + [C.c_int, C.c_double]
        L.zdo_pcg_seed.argtypes = [C.POINTER(Pcg), C.c_uint64]
        L.zdo_pcg_next.argtypes = [C.POINTER(Pcg)]
        L.zdo_pcg_advance.argtypes = [C.POINTER(Pcg), C.c_uint64, C.c_uint64]
        L.zdo_pcg_distance.argtypes = [C.POINTER(Pcg), C.POINTER(Pcg)]
        L.zdo_run.argtypes = [C.POINTER(Params), C.POINTER(Pk), C.c_void_p, C.c_int64, C.c_void_p,
                              C.c_void_p, C.c_void_p, C.POINTER(Stats)]
        L.zdo_mode_cube.argtypes = [C.POINTER(Params), C.POINTER(Pk), C.c_void_p, C.c_int64, C.c_void_p]
        L.zdo_mode_draw.argtypes = [C.POINTER(Params), C.POINTER(Pk), C.c_int, C.c_int, C.c_int,
                                    C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
        L.zdo_get_eigenmode.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int64,
                                        C.c_int, C.POINTER(C.c_double)]
        L.zdo_direct_sum.argtypes = [C.POINTER(Params), C.POINTER(Pk), C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
        L.zdo_fft_backend.restype = C.c_char_p
        L.zdo_pk_set_primordial.argtypes = [C.POINTER(Pk), C.c_double]
        L.zdo_infer_Tk.argtypes = [C.POINTER(Pk), C.c_double]
        L.zdo_infer_Tk.restype = C.c_double
        L.zdo_spline_build.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.zdo_spline_val.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]
        L.zdo_blockarray_roundtrip.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]
        _lib = L
    return _lib


def fft_backend():
    """'fftw3' when the host has libfftw3.so.3 (what the reference calls), else the oracle's built-in 'radix-2'"""
    return lib().zdo_fft_backend().decode()


def have_ref():
    return os.path.exists(_REF)


_ref = None


def ref():
    """reference pcg64 / SplineFunction object code (this container only)"""
    global _ref
    if _ref is None:
        R = C.CDLL(_REF)
        R.ref_pcg_distance.restype = C.c_uint64
        R.ref_pcg_distance.argtypes = [C.c_uint64] * 4
        R.ref_pcg_seed.argtypes = [C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        R.ref_pcg_draw.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int, C.c_void_p]
        R.ref_pcg_advance.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_uint64, C.c_uint64]
        R.ref_spline_val.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        R.ref_blockarray_roundtrip.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]
        _ref = R
    return _ref


def blockarray_input(ppd, numblock, narray):
    """deterministic z-stage slabs [yblock][yres][a][z][x]: every element carries its own coordinates
    (re = 1 + linear index over (y, a, z, x), im = -re - 0.25), so any misplaced skewer shows"""
    n = ppd * narray * ppd * ppd
    re = 1.0 + np.arange(n, dtype=np.float64)
    out = np.empty((n, 2), dtype=np.float64)
    out[:, 0] = re
    out[:, 1] = -re - 0.25
    return out


def blockarray_roundtrip(fn, ppd, numblock, narray, fill=-7.5):
    """fn = lib().zdo_blockarray_roundtrip or ref().ref_blockarray_roundtrip; returns (arr, slabs_out) as float64 [n, 2]"""
    inp = blockarray_input(ppd, numblock, narray)
    arr = np.zeros_like(inp)
    out = np.zeros_like(inp)
    rc = fn(ppd, numblock, narray, inp.ctypes.data, arr.ctypes.data, out.ctypes.data, fill)
    if rc:
        raise RuntimeError("blockarray roundtrip failed")
    return arr, out


def make_params(ppd, numblock=2, boxsize=720.0, seed=12346, k_cutoff=1.0, qPLT=0, qPLTrescale=0,
                PLT_target_z=0.0, z_initial=49.0, f_cluster=1.0, icformat="RVdoubleZel", qdensity=0,
                qoneslab=-1, qonemode=0, one_mode=(0, 0, 0), CornerModes=0, cpd=None, nthreads=0,
                f_NL=0.0, n_s=1.0, Omega_M=1.0, version=2):
    p = Params()
    p.ppd = ppd
    if version == 1 and k_cutoff != 1.0:  # src/parameters.cpp:129-141
        numblock = int(numblock * k_cutoff + .5)
    p.version = version
    p.numblock = numblock
    p.cpd = cpd if cpd is not None else ppd
    p.boxsize = boxsize
    # src/parameters.cpp:172-174
    p.separation = boxsize / ppd
    p.nyquist = np.pi / p.separation
    p.fundamental = 2.0 * np.pi / boxsize
    p.k_cutoff = k_cutoff
    p.qdensity = qdensity
    p.qoneslab = qoneslab
    p.seed = seed
    p.f_cluster = f_cluster
    p.qonemode = qonemode
    p.one_mode = (C.c_int * 3)(*one_mode)
    p.qPLT = qPLT
    p.qPLTrescale = qPLTrescale
    p.PLT_target_z = PLT_target_z
    p.z_initial = z_initial
    p.CornerModes = CornerModes
    p.icformat = ICFORMATS[icformat]
    p.nthreads = nthreads
    p.f_NL, p.n_s, p.Omega_M = f_NL, n_s, Omega_M
    return p


def pk_from_file(path, boxsize, Pk_scale=1.0, Pk_norm=8.0, Pk_sigma=0.0210839935761, Pk_sigma_ratio=0.0,
                 Pk_smooth=0.0, fix_to_mean=0):
    pk = Pk()
    rc = lib().zdo_pk_from_file(C.byref(pk), path.encode(), Pk_scale, Pk_norm, Pk_sigma, Pk_sigma_ratio,
                                Pk_smooth, fix_to_mean, boxsize)
    if rc:
        raise RuntimeError("zdo_pk_from_file failed")
    return pk


def pk_from_powerlaw(index, boxsize, Pk_norm=8.0, Pk_sigma=0.02, Pk_sigma_ratio=0.0, Pk_smooth=0.0,
                     fix_to_mean=0):
    pk = Pk()
    lib().zdo_pk_from_powerlaw(C.byref(pk), index, Pk_norm, Pk_sigma, Pk_sigma_ratio, Pk_smooth,
                               fix_to_mean, boxsize)
    return pk


def pk_tables(pk):
    n = pk.n
    return (np.ctypeslib.as_array(pk.x, (n,)).copy(), np.ctypeslib.as_array(pk.y, (n,)).copy(),
            np.ctypeslib.as_array(pk.y2, (n,)).copy())


def run(params, pk, eig=None, eig_ppd=0, want_planes=False, want_density=False):
    """returns dict(records=structured array [z,y,x], planes=[z,a,y,x] complex, stats)"""
    L = lib()
    n = int(params.ppd)
    fmt = [k for k, v in ICFORMATS.items() if v == params.icformat][0]
    dt = RECORD_DTYPES[fmt]
    na = L.zdo_narray(C.byref(params))
    # np.empty + fill: touch every page up front (first-touch faults are not part of the path's timers)
    rec = np.empty(n * n * n, dtype=dt) if params.qdensity != 2 else None
    if rec is not None:
        rec.view(np.uint8).fill(0)
    planes = np.empty((n, na, n, n), dtype=np.complex128) if want_planes else None
    if planes is not None:
        planes.fill(0)
    dens = np.empty(n * n * n, dtype=np.float32) if want_density else None
    if dens is not None:
        dens.fill(0)
    st = Stats()
    eigp = eig.ctypes.data if eig is not None else None
    rc = L.zdo_run(C.byref(params), C.byref(pk), eigp, eig_ppd,
                   rec.ctypes.data if rec is not None else None,
                   dens.ctypes.data if dens is not None else None,
                   planes.ctypes.data if planes is not None else None, C.byref(st))
    if rc:
        raise RuntimeError("zdo_run failed rc=%d" % rc)
    return dict(records=None if rec is None else rec.reshape(n, n, n), planes=planes,
                density=None if dens is None else dens.reshape(n, n, n),
                max_disp=np.array(list(st.max_disp)), density_variance=st.density_variance, stats=st)


def direct_sum(params, pk, sites, eig=None):
    """fields at the lattice sites [(z, y, x), ...] by direct summation over the modes (zdo_direct_sum): float64
    [nsites, 7] = qx, qy, qz, vx, vy, vz, density"""
    s = np.ascontiguousarray(sites, dtype=np.int32).reshape(-1, 3)
    out = np.zeros((s.shape[0], 7), dtype=np.float64)
    if eig is not None:
        eig = np.ascontiguousarray(eig, dtype=np.float64)
    rc = lib().zdo_direct_sum(C.byref(params), C.byref(pk), eig.ctypes.data if eig is not None else None,
                              eig.shape[0] if eig is not None else 0, s.shape[0], s.ctypes.data, out.ctypes.data)
    if rc:
        raise RuntimeError("zdo_direct_sum failed rc=%d" % rc)
    return out


def mode_cube(params, pk, eig=None, eig_ppd=0):
    L = lib()
    n = int(params.ppd)
    na = L.zdo_narray(C.byref(params))
    cube = np.zeros((na, n, n, n), dtype=np.complex128)  # [a][ky][kz][kx]
    rc = L.zdo_mode_cube(C.byref(params), C.byref(pk), eig.ctypes.data if eig is not None else None,
                         eig_ppd, cube.ctypes.data)
    if rc:
        raise RuntimeError("zdo_mode_cube failed")
    return cube


def synthetic_eigenmodes(ppd_e=128, seed=7, amp=0.03, singular=False):
    """Synthetic PLT eigenmode table in the reference's file layout (src/zeldovich.cpp:796-797,815,
    155-159): float64 [ppd_e][ppd_e][ppd_e/2+1][4] = (e_x,e_y,e_z,lambda), FFT order, |e|=1.
    eigmodes128 is absent from the reference mount, so PLT paths are exercised with this table:
    e = normalised(k_hat + amp min((k/k_Ny)^2, 1.5) M k_hat) with a fixed random 3x3 matrix M (entries in
    [-1, 1]) — within ~5 degrees of k_hat like real PLT eigenvectors, so k^2/(k.e) stays O(k) and the
    displacements O(1) — and lambda = 1 - 0.2 (k/k_Ny)^2.
    singular=True is the round-1 recipe (perturbation up to ~1.3 |k_hat|): e can come out almost
    perpendicular to k, k^2/(k.e) blows up at a handful of modes; kept as an edge-case input."""
    h = ppd_e // 2 + 1
    idx = np.arange(ppd_e)
    kfull = np.where(idx > ppd_e // 2, idx - ppd_e, idx).astype(np.float64)
    kx, ky, kz = kfull[:, None, None], kfull[None, :, None], np.arange(h, dtype=np.float64)[None, None, :]
    kn = ppd_e / 2.0
    k2 = kx * kx + ky * ky + kz * kz
    kk = np.sqrt(np.where(k2 > 0, k2, 1.0))
    q2 = k2 / (kn * kn)
    c = np.random.RandomState(seed).uniform(-1, 1, size=(3, 3))
    if singular:
        pert = [0.15 * q2 * (c[i, 0] * kx + c[i, 1] * ky + c[i, 2] * kz) / kn for i in range(3)]
    else:
        pert = [amp * np.minimum(q2, 1.5) * (c[i, 0] * kx + c[i, 1] * ky + c[i, 2] * kz) / kk for i in range(3)]
    e = [kx / kk + pert[0], ky / kk + pert[1], kz / kk + pert[2]]
    mag = np.sqrt(e[0] ** 2 + e[1] ** 2 + e[2] ** 2)
    mag = np.where(mag > 0, mag, 1.0)
    out = np.empty((ppd_e, ppd_e, h, 4), dtype=np.float64)
    for i in range(3):
        out[..., i] = e[i] / mag
    out[..., 3] = 1.0 - 0.2 * q2
    out[0, 0, 0, :3] = 0.0
    return np.ascontiguousarray(out)
