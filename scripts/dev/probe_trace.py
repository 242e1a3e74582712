"""Development probe: the reference's scene at a 512-spp slice on the development library with kz_debug_trace on: which passes a fresh replica's calls run, how each ran, what the
replica measured and kept (KzRenderOpts::shadowBeside / passHalves at their defaults)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
kz = importlib.import_module("nano-kazen_amd")
lib = kz.abi.load_dev_library()
lib.kz_debug_trace(1)
d = kz.scenes.load_npz(os.path.join(ROOT, "tests", "golden", "q1_default_m0_r0.5.npz"))
sc = kz.Scene(d, device=0, lib=lib)
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 512
sc.render(0, 64); sc.sync()
for i in range(8):
    t0 = time.perf_counter(); sc.render(0, spp); sc.sync(); dt = time.perf_counter() - t0
    info = sc.last_pass_info()
    print("call %d: %.4f s, %d passes, largest %d items, last pass ran as %d" % (i, dt, info["passes"], info["largestPassItems"], info["shadowBeside"]), file=sys.stderr, flush=True)
