import sys, time, importlib, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import oracle as O
kz = O.kz
for name, desc in [("cornell", kz.scenes.cornell_box(128,128,16)), ("sphere", kz.scenes.sphere_env(128,128,16)), ("cornell_pmj", kz.scenes.cornell_box(128,128,16,sampler="pmj02bn",seed=1))]:
    sc = kz.Scene(desc, device=0)
    t=time.time(); sc.render(); sc.sync(); t1=time.time()-t
    g = sc.rgb()
    o = O.OracleScene(desc); c = o.rgb(o.render())
    l2 = np.sqrt(np.mean((g-c)**2)); 
    print(name, "gpu %.3fs"%t1, "L2", l2, "max", np.abs(g-c).max(), "means", g.mean(), c.mean(), "kernel ms", sc.last_kernel_ms(), flush=True)
    sc.set_stats(True); sc.render(); print(" gpu stats", sc.stats()); print(" cpu stats", o.stats(), flush=True)
    # ray-level
    rng = np.random.default_rng(1)
    n=20000
    oo = rng.uniform(-0.9,0.9,(n,3)).astype(np.float32); dd = rng.normal(size=(n,3)).astype(np.float32); dd/=np.linalg.norm(dd,axis=1,keepdims=True)
    hg = sc.trace_rays(oo,dd,1e-3,np.inf); ob = O.OracleScene(desc, brute=True); hc = ob.trace_rays(oo,dd,1e-3,np.inf)
    same = (hg['mesh']==hc['mesh'])&(hg['prim']==hc['prim'])
    print(" rays: prim agree", same.mean(), "t maxdiff", np.nanmax(np.abs(np.where(np.isfinite(hc['t']), hg['t']-hc['t'],0))), "p maxdiff", np.abs(hg['p']-hc['p'])[same].max(), "n maxdiff", np.abs(hg['sh_n']-hc['sh_n'])[same].max(), flush=True)
