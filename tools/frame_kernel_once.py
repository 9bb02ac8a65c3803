#!/usr/bin/env python3
"""One 1080x960, 16-spp, depth-5 frame of the main scene with the frame kernel off (argument 0) or on (1): the program a
counter pass profiles."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

rt = entry.load_package()
gpu = rt.Scene.main(sky=rt.load_skybox(max_width=512), cam=(1080, 960)).upload(0)
gpu.set_option(rt.OPT_FRAME_KERNEL, int(sys.argv[1]))
aa = int(sys.argv[2]) if len(sys.argv) > 2 else 4
for _ in range(int(sys.argv[3]) if len(sys.argv) > 3 else 1):
    img, st = gpu.render(max_depth=5, shadow_rays=1, antialiasing=aa)
    print(st)
