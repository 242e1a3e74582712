import sys, importlib, numpy as np
sys.path.insert(0, '/root/repo')
kz = importlib.import_module("nano-kazen_amd")
d = kz.scenes.textured_scene(960, 576, 256, sampler="pmj02bn")
sc = kz.Scene(d, device=0)
sc.render()
kz.output.save_png("/root/repo/gpurun_out/textured_960", sc.srgb8())
print("ok", sc.last_kernel_ms())
