"""Shade-stage time per sampler plugin on the C4 geometry (how much of kz_wf_shade is the sampler?)."""
import sys, os, importlib
sys.path.insert(0, '/root/repo')
kz = importlib.import_module("nano-kazen_amd")
for smp in ("pmj02bn", "independent", "stratified", "correlated"):
    d = kz.scenes.random_triangles(1000000, 1920, 1080, 1024, sampler=smp)
    sc = kz.Scene(d, device=0)
    sc.render(32, 48); sc.sync(); sc.render(48, 64); sc.sync()
    print(smp, sc.last_kernel_ms(), sc.last_stage_ms(), flush=True)
    del sc
