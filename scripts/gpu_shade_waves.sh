#!/bin/bash
cd $GRAFT_REPO_ROOT
for w in 3 4 5; do
  KZ_EXTRA_HIPFLAGS="-DKZ_SHADE_WAVES=$w" sh nano-kazen_amd/csrc/build.sh > /dev/null 2>&1
  echo "== shade waves $w"
  python - <<'PY'
import sys, importlib
sys.path.insert(0,'/root/repo')
kz = importlib.import_module("nano-kazen_amd")
for name, d in (("C4", kz.scenes.random_triangles(1000000, 1920, 1080, 1024)), ("cornell", kz.scenes.cornell_box(1920,1080,16)), ("hero", kz.scenes.hero_scene(1920,1080,16,detail=2.0))):
    sc = kz.Scene(d, device=0)
    sc.render(0,16); sc.sync(); sc.render(0,16); sc.sync()
    print(name, "pass ms %.2f" % sc.last_kernel_ms(), flush=True)
    sc.close()
PY
done
