"""One frame of the bench scene with a librt355.so built with -DRT355_SHADE_TIMING (prints k_shade phase times of a few workgroups)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magr_ray_tracer_amd import scenes
from magr_ray_tracer_amd.renderer import Device
W, H = 1920, 1080
s, view = scenes.sponza_class(1.0)
sa = s.arrays()
cam = scenes.camera_for(view, W, H)
d = Device(W, H)
d.upload(sa)
d.seed_default()
d.render(cam, 1)
d.synchronize()
print("---- second frame")
d.render(cam, 1)
d.synchronize()
d.close()
