"""CPU: host-side logic of the product (no compute on a GPU): parameter file reader, PowerSpectrum
setup vs the oracle, eigenmode loader, stream-factor chooser, FFT engine index arithmetic (host
emulation of zd_fft.h) and the RNG jump maps."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, WMAP

EXAMPLE = """# An example zeldovich parameter file (same keys as the reference's example.par)
BoxSize = 720
CPD = 375
ICFormat = "RVZel"
InitialConditionsDirectory = "%(out)s"
InitialRedshift = 49
NP = 2097152
ZD_NumBlock = 4
ZD_PLT_filename = "%(eig)s"
ZD_PLT_target_z = 5
ZD_Pk_filename = "%(pk)s"
ZD_Pk_norm = 8.0
ZD_Pk_scale = 1.0
ZD_Pk_sigma = 0.0210839935761
ZD_Pk_smooth = 0.0
ZD_Seed = 12346
ZD_k_cutoff = 1.0
ZD_qPLT = %(plt)d
ZD_qPLT_rescale = 0
ZD_qPk_fix_to_mean = 0
ZD_Version = 2
ZD_f_NL = 0
"""


@pytest.fixture()
def zd():
    import zeldovich_plt_amd.api as api
    api.load_library()
    return api


def _write(tmp_path, text):
    f = tmp_path / "t.par"
    f.write_text(text)
    return str(f)


def test_stream_factor_for_any_even_ppd(zd):
    """PPD with other prime factors than 2 and 3 (zd_kernels_any.hip): reference arrays on one rank, R any divisor of PPD; a 2^a 3^b size falls back to the same path when its options need the reference arrays (ZD_qdensity = 2)"""
    L = zd.load_library()
    GB = 1 << 30

    def R(n, budget, nranks=1, **kw):
        return L.zd_choose_stream_factor(C.byref(zd.make_params(n, **kw)), nranks, int(budget))

    assert R(1000, 250 * GB) == 1          # 1000 * 2 * 1000 * 1024 * 16 B = 33 GB
    assert R(2000, 250 * GB) == 1 and R(2000, 200 * GB) == 2   # 259 GB at R = 1
    assert R(4000, 250 * GB) == 8          # 32 * 125: z lines of 500
    assert R(3000, 250 * GB) == 4 and R(3000, 150 * GB) == 6   # any divisor: the z-residue fold is a plain decimation
    assert R(5000, 250 * GB) == 20         # z lines of 250
    assert R(1001, 250 * GB) == -1         # odd PPD: the reference requires an even one too
    assert R(1000, 250 * GB, nranks=2) == -1
    assert R(1000, 250 * GB, qPLT=1) == 1 and R(1000, 250 * GB, qdensity=2) == 1
    assert R(96, 250 * GB) == 2 and R(96, 250 * GB, qdensity=1) == 2 and R(96, 250 * GB, qdensity=2) == 2   # composite kernels (ZA, also
    # with a density since round 4, density only since round 5: the six-field store) ...
    assert R(96, 250 * GB, qdensity=2, qPLT=1) == 2  # ... density only ignores ZD_qPLT, as the reference does (zeldovich.cpp:303,440)
    # PLT with ZD_qdensity = 1: a density-only pass at 2R in front of every PLT pass (plt_dens_split) — R and 2R must both be z lengths
    # the composite kernels transform, and the density planes of a pass need room
    assert R(96, 250 * GB, qdensity=1, qPLT=1) == 1
    assert R(3456, 270 * GB, qPLT=1) == 6 and R(3456, 270 * GB, qdensity=1, qPLT=1) == 8  # 6: no room beside the 260 GB store; 8 with 2R = 16: z lines of 216 = 8 * 27
    assert R(3456, 270 * GB, qdensity=1, qPLT=1, nranks=2) == -1  # several ranks of one group: no composite path, and the convolution path is one rank
    assert R(1000, 250 * GB, f_NL=1.0, n_s=0.96, Omega_M=0.3) == 1
    # radix-7 composite grids (round 4): 7168 = 1024 * 7 with z lines of 112 = 16 * 7; 3584 = 512 * 7, z lines of 448; 4320 = 32 * 135
    # has no z kernel of 16 * 135 (1080 threads): z lines of 432 = 16 * 27
    assert R(7168, 250 * GB) == 64 and R(3584, 250 * GB) == 8 and R(2688, 250 * GB) == 2 and R(4320, 250 * GB) == 10
    assert R(7168, 250 * GB, nranks=8) == 8 and R(6272, 250 * GB, qPLT=1) == 56
    assert R(8192, 250 * GB, qPLT=1, k_cutoff=2.0) > 0 and R(8192, 250 * GB, qPLT=1, qdensity=1) == -1   # four reference arrays at 8192: a row of the x pass is 2048 threads
    assert R(8640, 250 * GB) == 72 and R(8640, 250 * GB, qPLT=1) == -1   # 64 * 135: z lines of 120 = 8 * 15 (round 4: R = 80, lines of 108 = 4 * 27); beyond 8192 only the ZA field store



def test_pass_groups_for_jobs_on_the_convolution_kernels(zd):
    """a job that one rank runs as convolutions (PPD neither 2^a nor a composite size with the options the composite kernels have)
    is shared by several GPUs as pass groups only: the library picks a divisor of PPD that deals its passes out over them — up to four
    times the passes one GPU would need, or two per GPU — and refuses where no divisor does (it used to refuse all of them)"""
    L = zd.load_library()
    GB = 1 << 30

    def G(n, ngpu, budget=250 * GB, **kw):
        g, R = C.c_int32(), C.c_int32()
        rc = L.zd_choose_pass_groups(C.byref(zd.make_params(n, **kw)), ngpu, int(budget), C.byref(g), C.byref(R))
        return (g.value, R.value) if rc == 0 else None

    assert G(1000, 8) == (8, 8) and G(1000, 2) == (2, 2) and G(3000, 8) == (8, 8)
    assert G(5000, 8) == (8, 40)       # one GPU needs R = 20 (z lines of 250): 40 passes, five per GPU
    assert G(100, 2) == (2, 2) and G(100, 4) == (4, 4) and G(50, 2) == (2, 2)
    assert G(1250, 4) is None          # 2 * 5^4 has no divisor that is a multiple of 4
    assert G(192, 4, k_cutoff=2.0, corner_modes=1) == (4, 4)     # a composite size whose option needs the reference arrays
    assert G(192, 4, k_cutoff=2.0, corner_modes=1, stream_factor=2) == (1, 2)  # a given factor is kept (plan creation then says that this store runs on one rank)
    assert G(3456, 8) == (1, 2) and G(4096, 8) == (8, 16)        # the composite / power-of-two choices are what they were

def test_params_from_file(zd, tmp_path):
    par = _write(tmp_path, EXAMPLE % dict(out=tmp_path / "ic", eig="./eigmodes128", pk=WMAP, plt=1))
    p, s = zd.params_from_file(par)
    assert p.ppd == 128 and p.numblock == 4 and p.cpd == 375 and p.seed == 12346
    assert p.qPLT == 1 and p.qPLTrescale == 0 and p.PLT_target_z == 5 and p.z_initial == 49
    assert p.icformat == zd.ICFORMATS["RVZel"] and s.ICFormat == b"RVZel"
    assert s.Pk_filename.decode() == WMAP and s.version == 2 and s.np == 2097152
    # derived quantities: src/parameters.cpp:172-174
    assert p.fundamental == 2.0 * np.pi / 720 and p.nyquist == np.pi / (720 / 128)
    assert p.k_cutoff == 1.0 and p.f_cluster == 1.0 and p.qoneslab == -1  # defaults :13-44


@pytest.mark.parametrize("mutation,ok", [
    (("ZD_Version = 2", ""), False),                    # ZD_Version must be given
    (("ZD_Version = 2", "ZD_Version = 3"), False),      # parameters.cpp:111
    (("ZD_Version = 2", "ZD_Version = 1\nZD_k_cutoff = 2.0"), True),  # legacy streams: NumBlock scaled by k_cutoff (:129-141)
    (("NP = 2097152", "NP = 2097153"), False),          # ppd^3 != NP
    (("BoxSize = 720", ""), False),                     # MUST_DEFINE
    (("ZD_Pk_sigma = 0.0210839935761", "ZD_Pk_sigma = 0.02\nZD_Pk_sigma_ratio = 1.0"), False),
    (("ZD_NumBlock = 4", "ZD_NumBlock = 3"), False),    # even divisor (block_array.cpp:38-40)
    (("ICFormat = \"RVZel\"", "ICFormat = \"Zeldovich\""), False),  # PLT needs an RV format
    (("ZD_k_cutoff = 1.0", "ZD_k_cutoff = 0.5"), False),
    (("ZD_Seed = 12346", "ZD_Seed = -7   # negative seeds sign-extend"), True),
    (("ZD_f_NL = 0", "ZD_f_NL = 50\nZD_n_s = 0.965\nOmega_M = 0.3"), True),
    (("NP = 2097152", "NP = 2.097152D6"), True),        # Fortran exponent
])
def test_params_validation(zd, tmp_path, mutation, ok):
    text = (EXAMPLE % dict(out=tmp_path / "ic", eig="./eigmodes128", pk=WMAP, plt=1)).replace(*mutation)
    par = _write(tmp_path, text)
    if ok:
        p, s = zd.params_from_file(par)
        if "Seed" in mutation[1]:
            assert p.seed == -7
        if "ZD_Version = 1" in mutation[1]:
            assert p.version == 1 and p.numblock == 8
        else:
            assert p.version == 2 and p.numblock == 4
    else:
        with pytest.raises(ValueError):
            zd.params_from_file(par)


def test_power_spectrum_table_order_quirk_matches_oracle(zd, oracle, tmp_path):
    """a P(k) file whose rows are NOT ascending: the reference's table sort compares one slot to the right of the element it
    moves (include/spline_function.h:77-104), so such a table comes out in the reference's own order — host library and
    oracle must agree bit for bit there too (ascending files pass through unchanged)"""
    t = np.loadtxt(WMAP)
    idx = np.random.default_rng(3).permutation(len(t))[:10]
    t[idx] = t[idx[::-1]]
    path = tmp_path / "shuffled.pow"
    np.savetxt(path, t)
    ps = zd.PowerSpectrum.from_file(str(path), 720.0)
    opk = oracle.pk_from_file(str(path), 720.0)
    x, y, y2 = ps.tables()
    ox, oy, oy2 = oracle.pk_tables(opk)
    assert np.array_equal(x, ox) and np.array_equal(y, oy) and np.array_equal(y2, oy2, equal_nan=True)
    assert ps.pk.normalization == opk.normalization


def test_power_spectrum_matches_oracle(zd, oracle):
    ps = zd.PowerSpectrum.from_file(WMAP, 720.0)
    opk = oracle.pk_from_file(WMAP, 720.0)
    x, y, y2 = ps.tables()
    ox, oy, oy2 = oracle.pk_tables(opk)
    assert np.array_equal(x, ox) and np.array_equal(y, oy)
    assert np.max(np.abs(y2 - oy2)) <= 1e-15 * np.max(np.abs(oy2))
    assert abs(ps.pk.normalization - opk.normalization) <= 1e-15 * opk.normalization
    L = oracle.lib()
    for k in [1e-3, 0.0123, 0.5, 3.0, 7.5]:
        a, b = ps.power(k), L.zdo_power(C.byref(opk), k)
        assert abs(a - b) <= 1e-15 * b
    assert ps.power(0.0) == 0.0 and ps.power(-1.0) == 0.0
    # sigma(R) after normalisation reproduces the target (x boxsize^1.5 undoes the 1/V)
    assert abs(ps.sigmaR(8.0) * 720.0 ** 1.5 - 0.0210839935761) < 1e-6 * 0.0210839935761
    pl = zd.PowerSpectrum.from_powerlaw(-1.5, 720.0)
    opl = oracle.pk_from_powerlaw(-1.5, 720.0)
    assert abs(pl.pk.normalization - opl.normalization) <= 1e-15 * opl.normalization
    assert abs(pl.power(0.3) - L.zdo_power(C.byref(opl), 0.3)) <= 1e-15 * pl.power(0.3)


def test_eigenmode_loader(zd, oracle, tmp_path):
    eig = oracle.synthetic_eigenmodes(8)
    f = tmp_path / "eig8"
    with open(f, "wb") as fh:
        fh.write(np.int32(8).tobytes())
        fh.write(eig.tobytes())
    L = zd.load_library()
    ptr, ppd = C.c_void_p(), C.c_int64()
    assert L.zd_load_eigmodes(os.fsencode(str(f)), C.byref(ptr), C.byref(ppd)) == 0
    got = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_double)), (eig.size,)).copy()
    L.zd_free(ptr)
    assert ppd.value == 8 and np.array_equal(got, eig.ravel())
    with open(f, "ab") as fh:
        fh.write(b"x")  # size mismatch must be rejected (zeldovich.cpp:817-822)
    assert L.zd_load_eigmodes(os.fsencode(str(f)), C.byref(ptr), C.byref(ppd)) != 0
    assert L.zd_load_eigmodes(b"/nonexistent/eig", C.byref(ptr), C.byref(ppd)) != 0


def test_stream_factor_chooser(zd):
    L = zd.load_library()
    GB = 1 << 30
    # ZA without ZD_qdensity: field store, two z-residues per pass (4 half-space potentials, zero columns not stored,
    # 384-B row pad + the y->x ring): ~111 GB at R = 2, which is preferred over R = 1
    p = zd.make_params(2048)
    assert L.zd_choose_stream_factor(C.byref(p), 1, 300 * GB) == 2
    assert L.zd_choose_stream_factor(C.byref(p), 1, 200 * GB) == 2
    assert L.zd_choose_stream_factor(C.byref(p), 1, 100 * GB) == 4
    assert L.zd_choose_stream_factor(C.byref(p), 8, 100 * GB) == 2   # one store per rank + the exchange ring
    pd = zd.make_params(2048, qdensity=1)  # density wanted: the reference's 2 arrays, 259 GiB at R = 1
    assert L.zd_choose_stream_factor(C.byref(pd), 1, 300 * GB) == 1
    assert L.zd_choose_stream_factor(C.byref(pd), 1, 200 * GB) == 2
    p4 = zd.make_params(4096)
    assert L.zd_choose_stream_factor(C.byref(p4), 1, 200 * GB) == 16
    assert L.zd_choose_stream_factor(C.byref(p4), 1, 260 * GB) == 8    # 4 passes on one 288 GB GPU
    assert L.zd_choose_stream_factor(C.byref(p4), 8, 200 * GB) == 2    # 8 GPUs: ONE pass (round 1: R = 4, send + receive stores)
    pr = zd.make_params(4096, store_mode="packed")  # round-1 packing: 3 arrays with Hermitian twins, 8 passes
    assert L.zd_choose_stream_factor(C.byref(pr), 1, 260 * GB) == 16
    assert L.zd_choose_stream_factor(C.byref(p4), 1, 1 * GB) == -1
    # composite grid (2^7 3^3): four potentials per pass at R = 4; with ZD_qdensity = 1 the six-field store (round 4) needs R = 6 —
    # still the composite kernels, not the any-divisor factors of the convolution path (reference arrays: 2 x 3456^3 x 16 B = 1.3 TB / R)
    pc, pcd = zd.make_params(3456), zd.make_params(3456, qdensity=1)
    assert L.zd_choose_stream_factor(C.byref(pc), 1, 256 * GB) == 4
    assert L.zd_choose_stream_factor(C.byref(pcd), 1, 256 * GB) == 6
    # ZD_qdensity = 2 (density only): the same six-field store since round 5 (only its density array is built); PLT with a density stays
    # on the convolution path (four reference arrays, any divisor)
    pc2 = zd.make_params(3456, qdensity=2)
    assert L.zd_choose_stream_factor(C.byref(pc2), 1, 256 * GB) == 6
    pc3 = zd.make_params(3456, qdensity=1, qPLT=1)
    assert L.zd_choose_stream_factor(C.byref(pc3), 1, 256 * GB) not in (-1, 4, 6)


# ---- host emulation of the device FFT engine -------------------------------------------------------
@pytest.fixture(scope="module")
def emul():
    src = os.path.join(ROOT, "tests", "host_emul", "emul_fft.cpp")
    out = os.path.join(ROOT, "tests", "host_emul", "_build", "libemul_fft.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not os.path.exists(out) or os.path.getmtime(out) < max(
            os.path.getmtime(src), os.path.getmtime(os.path.join(ROOT, "zeldovich_plt_amd", "csrc", "zd_fft.h"))):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", src, "-o", out])
    return C.CDLL(out)


@pytest.mark.parametrize("n,e,w", [(16, 16, 2), (32, 16, 2), (64, 16, 4), (128, 16, 4), (256, 16, 2), (512, 16, 2),
                                   (1024, 16, 2), (2048, 16, 2), (4096, 16, 1), (64, 8, 4), (512, 8, 2),
                                   (128, 4, 2), (32, 2, 2), (8, 4, 3)])
@pytest.mark.parametrize("line", [0, 1])
def test_fft_engine_host_emulation(emul, n, e, w, line):
    """the exact pass / exchange index arithmetic the kernels run, executed thread by thread on the CPU"""
    rng = np.random.default_rng(n * 31 + e + line)
    x = rng.standard_normal((w, n)) + 1j * rng.standard_normal((w, n))
    out = np.zeros_like(x)
    assert emul.emul_fft(n, e, w, line, x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)) == 0
    ref = np.fft.ifft(x, axis=1) * n
    assert np.abs(out - ref).max() / np.abs(ref).max() < 5e-15


def test_parseheader_grammar_features(zd, tmp_path):
    """include files, `##` block comments, backslash continuation, logical keywords, declarations without `=`
    (subprojects/ParseHeader/src/phScanner.ll) — what Abacus-style parameter files actually use"""
    inc = tmp_path / "common.par"
    inc.write_text("BoxSize = 720   # from the include file\nCPD = 375\nZD_Pk_scale = 1.0\n")
    text = (EXAMPLE % dict(out=tmp_path / "ic", eig="./eigmodes128", pk=WMAP, plt=0))
    text = text.replace("BoxSize = 720\n", "").replace("CPD = 375\n", "").replace("ZD_Pk_scale = 1.0\n", "")
    text = ('include "%s"\n' % inc) + text
    text += "##\nNP = 5   this whole block is a comment\n##\n"
    text += "ZD_qonemode = true\nZD_one_mode = 1 \\\n   2 \\\n  -3\n"
    text += "vcounter nrows\nvector a b\n1 2\n3 4\nmapvar m a\n"
    par = tmp_path / "g.par"
    par.write_text(text)
    p, s = zd.params_from_file(str(par))
    assert p.boxsize == 720 and p.cpd == 375 and p.ppd == 128
    assert p.qonemode == 1 and list(p.one_mode) == [1, 2, -3]
    with pytest.raises(ValueError):
        bad = tmp_path / "bad.par"
        bad.write_text('include "/nonexistent/file.par"\n' + text)
        zd.params_from_file(str(bad))


def test_parseheader_vector_vcounter_mapvar(zd, tmp_path):
    """`vcounter` / `vector` blocks and `mapvar` aliases (subprojects/ParseHeader/src/phParser.yy:74-166,
    phDriver.cc:433-531) are parsed, not skipped: a vector block can fill an installed vector (ZD_one_mode), a mapvar
    alias assigns its base variable, tables of variables this program does not know are accepted, and the grammar's
    errors are errors"""
    base = EXAMPLE % dict(out=tmp_path / "ic", eig="./eigmodes128", pk=WMAP, plt=0)
    text = base.replace("ZD_Seed = 12346\n", "")
    text += "mapvar ZD_Seed MasterSeed LegacySeed\nMasterSeed = 777      # assigns ZD_Seed\n"
    text += "ZD_qonemode = 1\nvcounter nmode\nvector ZD_one_mode\n3\n5\n-7\n"
    text += "vcounter nout\nvector OutputRedshift OutputLabel   # an Abacus-style table of another program\n3.0 'z3'\n1.5 \"z1.5\"\n0.0 z0\nnfinal = 3\n"
    par = tmp_path / "v.par"
    par.write_text(text)
    p, s = zd.params_from_file(str(par))
    assert p.seed == 777
    assert p.qonemode == 1 and list(p.one_mode) == [3, 5, -7]

    def fails(extra):
        bad = tmp_path / "bad.par"
        bad.write_text(base + extra)
        with pytest.raises(ValueError):
            zd.params_from_file(str(bad))

    fails("vector a b\n1 2\n")                                  # vector before any vcounter (phDriver.cc:507-509)
    fails("vcounter n\nvector a b\n1 2 3\n")                    # row length != number of variables
    fails("mapvar ZD_Seed s1\nmapvar BoxSize s1\n")              # already mapped (phDriver.cc:442-446)
    fails("BoxSize 720\n")                                       # statement without '=' (phParser.yy:95-99)
    fails("vcounter n\nvector a\n1\n\n2\n")                    # a blank line ends the block: the next row is a syntax error


@pytest.mark.parametrize("how", ["segv", "abort", "term"])
def test_bench_last_words_handler(how):
    """bench.py's arm_last_words: a process that dies inside a C call (a fault, abort(), the launcher's SIGTERM while it sits in a
    blocking call with the GIL released) still writes its line to stdout — and ends with status 128 + signal, not 0: a launcher must
    see a faulted rank as failed (VERDICT r4 #3a)"""
    import subprocess
    import sys
    import time
    code = ("import bench, ctypes\n"
            "keep = bench.arm_last_words(lambda s: 'LAST WORDS %d' % s)\n"
            + {"segv": "ctypes.string_at(0)\n", "abort": "ctypes.CDLL(None).abort()\n",
               "term": "print('ready', flush=True)\nctypes.CDLL(None).sleep(60)\n"}[how])
    p = subprocess.Popen([sys.executable, "-c", code], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    if how == "term":
        assert p.stdout.readline().strip() == "ready"
        time.sleep(0.3)
        p.terminate()
    out, _ = p.communicate(timeout=60)
    import signal
    want = {"segv": signal.SIGSEGV, "abort": signal.SIGABRT, "term": signal.SIGTERM}[how]
    assert ("LAST WORDS %d" % int(want)) in out and p.returncode == 128 + int(want)
