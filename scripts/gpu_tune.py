import sys, os, importlib, itertools
sys.path.insert(0,'/root/repo')
kz = importlib.import_module("nano-kazen_amd")
d = kz.scenes.random_triangles(1000000, 1920, 1080, 1024)
sc = kz.Scene(d, device=0)
def run(**env):
    for k,v in env.items(): os.environ[k]=str(v)
    sc.render(32,48); sc.sync(); sc.render(48,64); sc.sync()
    return sc.last_kernel_ms()
base = dict(KZ_TUNE_REFILL=44, KZ_TUNE_POSTPONE=20, KZ_TUNE_BATCH=256, KZ_TUNE_TRAV_BLOCKS=8, KZ_TUNE_SHADE_BLOCKS=6)
print("base", run(**base), flush=True)
for k, vals in (("KZ_TUNE_REFILL",(1,16,32,40,48,56,64,65)), ("KZ_TUNE_POSTPONE",(0,8,16,24,32,48)), ("KZ_TUNE_BATCH",(64,128,512,1024)), ("KZ_TUNE_TRAV_BLOCKS",(4,6,10,16,32)), ("KZ_TUNE_SHADE_BLOCKS",(2,4,8,16))):
    for v in vals:
        e = dict(base); e[k]=v
        print(k, v, "%.2f" % run(**e), flush=True)
