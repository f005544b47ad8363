"""GPU: the drop-in command line `zeldovich <param_file>` (reference src/zeldovich.cpp:848-1032):
ic_{z*CPD/PPD} files in the reference's record formats, plane order inside a file = increasing z even
when planes are produced in residue order, optional density file, usage/exit codes."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, WMAP

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "zeldovich_plt_amd", "csrc", "build", "zeldovich")

PAR = """BoxSize = 720
CPD = %(cpd)d
ICFormat = "%(fmt)s"
InitialConditionsDirectory = "%(out)s"
InitialRedshift = 49
NP = %(np)d
ZD_NumBlock = 2
ZD_Pk_filename = "%(pk)s"
ZD_Pk_norm = 8.0
ZD_Pk_scale = 1.0
ZD_Pk_sigma = 0.0210839935761
ZD_Pk_smooth = 0.0
ZD_Seed = 12346
ZD_Version = 2
ZD_qdensity = %(qd)d
ZD_StreamFactor = %(R)d
"""


def test_usage_and_bad_file():
    r = subprocess.run([EXE], capture_output=True, text=True)
    assert r.returncode == 1 and "Usage:" in r.stderr
    r = subprocess.run([EXE, "/nonexistent.par"], capture_output=True, text=True)
    assert r.returncode == 1


@pytest.mark.parametrize("fmt,R,qd,ngpu", [("RVdoubleZel", 1, 0, 1), ("RVZel", 2, 1, 1), ("ZelSimple", 2, 0, 1),
                                          ("Zeldovich", 2, 0, 1), ("RVZel", 2, 0, 2), ("RVdoubleZel", 4, 0, 4),
                                          ("RVZel", 2, 1, -70),    # ngpu < 0: one GPU, NP = 70^3 (2 * 5 * 7: the any-PPD kernels)
                                          ("RVZel", 2, 1, -96)])   # NP = 96^3 (2^5 3) with a density file: the six-field store of the composite kernels
def test_cli_writes_reference_files(tmp_path, oracle, fmt, R, qd, ngpu):
    """ngpu > 1: `ZD_NumGPU` in the parameter file — one host thread per rank inside the library; on this one-GPU box the
    ranks share the device (see test_native_multi_gpu_driver)"""
    n, cpd = (64, 5) if ngpu == 1 else (128, 7)
    if ngpu < 0:
        n, cpd, ngpu = -ngpu, 9, 1
    out = tmp_path / "ic"
    out.mkdir()
    (out / "ic_99").write_bytes(b"stale")       # SetupOutputDir removes ic_* and zeldovich.* (output.cpp:236-251)
    (out / "keep.txt").write_text("keep")
    par = tmp_path / "t.par"
    par.write_text(PAR % dict(cpd=cpd, fmt=fmt, out=out, np=n ** 3, pk=WMAP, qd=qd, R=R)
                   + ("ZD_NumGPU = %d\nZD_ExchangePlanes = 3\nZD_PassGroups = 1\n" % ngpu if ngpu > 1 else ""))
    r = subprocess.run([EXE, str(par)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Mpart/sec" in r.stderr and "maximum component-wise displacements" in r.stderr
    assert not (out / "ic_99").exists() and (out / "keep.txt").exists()
    pk = oracle.pk_from_file(WMAP, 720.0)
    ref = oracle.run(oracle.make_params(n, numblock=2, icformat=fmt, cpd=cpd, qdensity=qd), pk, want_density=bool(qd))
    dt = oracle.RECORD_DTYPES[fmt]
    files = sorted(int(f.name[3:]) for f in out.iterdir() if f.name.startswith("ic_"))
    assert files == sorted(set(z * cpd // n for z in range(n)))
    for f in files:
        zs = [z for z in range(n) if z * cpd // n == f]
        got = np.fromfile(out / ("ic_%d" % f), dtype=dt).reshape(len(zs), n, n)
        want = ref["records"][zs]
        if "ijk" in dt.names:
            assert np.array_equal(got["ijk"], want["ijk"])
        tol = 1e-10 if dt["d"].base == np.float64 else 1e-6
        assert np.abs(got["d"] - want["d"]).max() <= tol * np.abs(want["d"]).max()
        if "v" in dt.names:
            assert np.abs(got["v"] - want["v"]).max() <= tol * np.abs(want["v"]).max()
    if qd:
        dens = np.fromfile(out / ("density%d" % n), dtype=np.float32).reshape(n, n, n)
        assert np.abs(dens - ref["density"]).max() <= 1e-6 * np.abs(ref["density"]).max()


def test_cli_version1(tmp_path, oracle):
    """`ZD_Version = 1` parameter files (legacy phases; NumBlock scaled by ZD_k_cutoff as src/parameters.cpp:129-141)"""
    n, cpd = 64, 5
    out = tmp_path / "ic"
    out.mkdir()
    par = tmp_path / "v1.par"
    text = PAR % dict(cpd=cpd, fmt="RVdoubleZel", out=out, np=n ** 3, pk=WMAP, qd=0, R=2)
    par.write_text(text.replace("ZD_Version = 2", "ZD_Version = 1").replace("ZD_NumBlock = 2", "ZD_NumBlock = 4") + "ZD_k_cutoff = 2.0\n")
    r = subprocess.run([EXE, str(par)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "ZD_Version = 1" in r.stderr and "NumBlock=8" in r.stderr
    pk = oracle.pk_from_file(WMAP, 720.0)
    ref = oracle.run(oracle.make_params(n, numblock=4, icformat="RVdoubleZel", cpd=cpd, k_cutoff=2.0, version=1), pk)
    dt = oracle.RECORD_DTYPES["RVdoubleZel"]
    for f in sorted(set(z * cpd // n for z in range(n))):
        zs = [z for z in range(n) if z * cpd // n == f]
        got = np.fromfile(out / ("ic_%d" % f), dtype=dt).reshape(len(zs), n, n)
        want = ref["records"][zs]
        assert np.abs(got["d"] - want["d"]).max() <= 1e-10 * np.abs(want["d"]).max()


def test_cli_plt_on_a_composite_grid(tmp_path, oracle):
    """the production Abacus shape in small: NP = 96^3 (2^5 3) with ZD_qPLT + rescale from an eigenmode FILE in the
    reference's layout (int32 ppd + ppd^2 (ppd/2+1) 4 doubles, src/zeldovich.cpp:794-830), through `zeldovich <param_file>`"""
    n, cpd = 96, 7
    out = tmp_path / "ic"
    out.mkdir()
    eig = oracle.synthetic_eigenmodes(32)
    eigfile = tmp_path / "eigmodes32"
    with open(eigfile, "wb") as f:
        f.write(np.int32(32).tobytes())
        f.write(np.ascontiguousarray(eig, dtype=np.float64).tobytes())
    par = tmp_path / "plt.par"
    par.write_text(PAR % dict(cpd=cpd, fmt="RVZel", out=out, np=n ** 3, pk=WMAP, qd=0, R=2)
                   + 'ZD_qPLT = 1\nZD_qPLT_rescale = 1\nZD_PLT_target_z = 5.0\nZD_PLT_filename = "%s"\nZD_f_cluster = 0.97\n' % eigfile)
    r = subprocess.run([EXE, str(par)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    pk = oracle.pk_from_file(WMAP, 720.0)
    ref = oracle.run(oracle.make_params(n, numblock=2, icformat="RVZel", cpd=cpd, qPLT=1, qPLTrescale=1, PLT_target_z=5.0,
                                        f_cluster=0.97), pk, eig=eig, eig_ppd=32)
    dt = oracle.RECORD_DTYPES["RVZel"]
    for f in sorted(set(z * cpd // n for z in range(n))):
        zs = [z for z in range(n) if z * cpd // n == f]
        got = np.fromfile(out / ("ic_%d" % f), dtype=dt).reshape(len(zs), n, n)
        want = ref["records"][zs]
        assert np.array_equal(got["ijk"], want["ijk"])
        assert np.abs(got["d"] - want["d"]).max() <= 1e-6 * np.abs(want["d"]).max()
        assert np.abs(got["v"] - want["v"]).max() <= 1e-6 * np.abs(want["v"]).max()


# BASELINE config 1 literally: the keys and values of the reference's example.par (CPD 375, NP = 128^3, ZD_NumBlock 4, RVZel,
# relative file names resolved in the working directory), with ZD_qPLT = 0 — the CPU-runnable case of BASELINE.json — and as
# shipped, ZD_qPLT = 1 with ./eigmodes128 (absent from the reference mount: a synthetic table in its layout, 128^3 cells = the
# exact lookup of src/zeldovich.cpp:161-170)
EXAMPLE_PAR = [("BoxSize", "720"), ("CPD", "375"), ("ICFormat", '"RVZel"'), ("InitialConditionsDirectory", '"./ic_out"'),
               ("InitialRedshift", "49"), ("NP", "2097152"), ("ZD_NumBlock", "4"), ("ZD_PLT_filename", '"./eigmodes128"'),
               ("ZD_PLT_target_z", "5"), ("ZD_Pk_filename", '"wmap1new.pow"'), ("ZD_Pk_norm", "8.0"), ("ZD_Pk_scale", "1.0"),
               ("ZD_Pk_sigma", "0.0210839935761"), ("ZD_Pk_smooth", "0.0"), ("ZD_Seed", "12346"), ("ZD_k_cutoff", "1.0"),
               ("ZD_qPLT", "1"), ("ZD_qPLT_rescale", "0"), ("ZD_qPk_fix_to_mean", "0"), ("ZD_Version", "2"), ("ZD_f_NL", "0")]


@pytest.mark.parametrize("qplt", [0, 1])
def test_cli_example_par_literally(tmp_path, oracle, qplt):
    import shutil
    n, cpd = 128, 375
    shutil.copy(WMAP, tmp_path / "wmap1new.pow")
    eig = oracle.synthetic_eigenmodes(128)
    with open(tmp_path / "eigmodes128", "wb") as f:
        f.write(np.int32(128).tobytes())
        f.write(np.ascontiguousarray(eig, dtype=np.float64).tobytes())
    (tmp_path / "example.par").write_text("# An example zeldovich parameter file.\n\n" + "".join(
        "%s = %s\n" % (k, str(qplt) if k == "ZD_qPLT" else v) for k, v in EXAMPLE_PAR))
    r = subprocess.run([EXE, "example.par"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    pk = oracle.pk_from_file(WMAP, 720.0)
    kw = dict(qPLT=1, qPLTrescale=0, PLT_target_z=5.0) if qplt else {}
    ref = oracle.run(oracle.make_params(n, numblock=4, icformat="RVZel", cpd=cpd, **kw), pk, eig=eig if qplt else None,
                     eig_ppd=128 if qplt else 0)
    dt = oracle.RECORD_DTYPES["RVZel"]
    out = tmp_path / "ic_out"
    files = sorted(int(f.name[3:]) for f in out.iterdir() if f.name.startswith("ic_"))
    assert files == sorted(set(z * cpd // n for z in range(n))) and len(files) == 128  # CPD > PPD: one plane per file
    scale_d, scale_v = np.abs(ref["records"]["d"]).max(), np.abs(ref["records"]["v"]).max()
    for f in files:
        zs = [z for z in range(n) if z * cpd // n == f]
        got = np.fromfile(out / ("ic_%d" % f), dtype=dt).reshape(len(zs), n, n)
        want = ref["records"][zs]
        assert np.array_equal(got["ijk"], want["ijk"])
        assert np.abs(got["d"] - want["d"]).max() <= 1e-6 * scale_d and np.abs(got["v"] - want["v"]).max() <= 1e-6 * scale_v
    assert "maximum component-wise displacements" in r.stderr
