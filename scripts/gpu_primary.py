"""Primary rays only (maxDepth 1 keeps one closest-hit query per sample + its shadow ray): counters and stage times on C4."""
import sys, os, importlib
os.environ["KZ_DUAL_STREAM"] = "0"
sys.path.insert(0, '/root/repo')
kz = importlib.import_module("nano-kazen_amd")
d = kz.scenes.random_triangles(1000000, 1920, 1080, 1024)
d.integrator["maxDepth"] = 1
sc = kz.Scene(d, device=0)
sc.render(0, 64); sc.sync(); sc.render(64, 128); sc.sync()
print("stages", sc.last_stage_ms(), flush=True)
sc.set_stats(True); sc.stats(reset=True)
sc.render(64, 128); sc.sync()
st = sc.stats(reset=True)
n = st["samples"]
print({k: round(v / n, 3) for k, v in st.items()})
