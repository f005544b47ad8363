"""s/step of 5-smooth PPDs on the composite-transform kernels (round 3: radix-5 stages), one GPU, ZA RVZel, automatic stream factor"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zeldovich_plt_amd.api as zd
ps = zd.PowerSpectrum.from_file(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "wmap1new.pow"), 720.0)
for n in [int(v) for v in (sys.argv[1:] or ["1280", "2560", "3200", "3840", "4000", "5120"])]:
    zd.generate(zd.make_params(n, icformat="RVZel"), ps, collect=False) if n <= 1280 else None  # warm-up for the small ones
    t = time.time()
    a = zd.generate(zd.make_params(n, icformat="RVZel", profile=1), ps, collect=False)
    print("PPD", n, "R", a["stream_factor"], "sec %.2f" % a["seconds_total"], "Gp/s %.2f" % (n ** 3 / a["seconds_total"] / 1e9),
          {k: round(v) for k, v in a["kernel_ms"].items()}, "wall %.1f" % (time.time() - t), flush=True)
