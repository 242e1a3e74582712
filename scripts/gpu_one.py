import sys, importlib
sys.path.insert(0,'/root/repo')
kz = importlib.import_module("nano-kazen_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "cornell"
d = {"cornell": lambda: kz.scenes.cornell_box(1920,1080,16), "hero": lambda: kz.scenes.hero_scene(1920,1080,16,detail=2.0),
     "c4": lambda: kz.scenes.random_triangles(1000000,1920,1080,1024), "mats": lambda: kz.scenes.materials_scene(1920,1080,16)}[name]()
sc = kz.Scene(d, device=0)
for i in range(3):
    sc.render(0,16); sc.sync()
print(name, "pass ms", sc.last_kernel_ms())
