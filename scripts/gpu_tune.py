import sys, os, importlib, itertools
sys.path.insert(0,'/root/repo')
kz = importlib.import_module("nano-kazen_amd")
d = kz.scenes.random_triangles(1000000, 1920, 1080, 1024)
sc = kz.Scene(d, device=0)
def run(**env):
    for k,v in env.items(): os.environ[k]=str(v)
    sc.render(32,48); sc.sync(); sc.render(48,64); sc.sync()
    return sc.last_kernel_ms()
base = dict(KZ_TUNE_REFILL=40, KZ_TUNE_POSTPONE=20, KZ_TUNE_BATCH=128, KZ_TUNE_TRAV_BLOCKS=8, KZ_TUNE_SHADE_BLOCKS=6, KZ_TUNE_LDS_STACK=16)
print("base", run(**base), flush=True)
for k, vals in (("KZ_TUNE_REFILL",(24,32,48,56,64)), ("KZ_TUNE_POSTPONE",(4,8,12,16,28,36)), ("KZ_TUNE_BATCH",(64,256)), ("KZ_TUNE_TRAV_BLOCKS",(6,7,10,12)), ("KZ_TUNE_LDS_STACK",(8,12,20,24)), ("KZ_TUNE_SHADE_BLOCKS",(3,4,8,12))):
    for v in vals:
        e = dict(base); e[k]=v
        print(k, v, "%.2f" % run(**e), flush=True)
