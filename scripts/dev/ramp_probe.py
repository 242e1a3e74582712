"""Development probe: the pass sizes the default schedule earns call by call (kz_render.hip: `earned`), with the library's trace on stderr.
    python scripts/dev/ramp_probe.py [--hold]     --hold: another scene holds a 2^30-item context meanwhile"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
kz = importlib.import_module("nano-kazen_amd")
lib = kz.abi.load_library()
lib.kz_debug_trace(1)
desc = kz.scenes.random_triangles(1000000, 1920, 1080, 1024, sampler="pmj02bn", seed=1)
hold = None
if "--hold" in sys.argv:
    hold = kz.Scene(desc, device=0)
    hold.render(0, 512, pass_items=1 << 30, passes_in_flight=1)
    hold.sync()
    print("hold:", hold.last_pass_info(), file=sys.stderr)
sc = kz.Scene(desc, device=0)
for k in range(5):
    sc.render(0, 512)
    sc.sync()
    print("call %d:" % k, sc.last_pass_info(), sc.last_grow_note(), file=sys.stderr)
