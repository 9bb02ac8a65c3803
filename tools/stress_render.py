import sys, os, itertools
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import __graft_entry__ as entry
rt = entry.load_package()
import oracle_lib as ol
sky = rt.load_skybox(max_width=512)
cam = (150, 110)
scenes = {"main": (rt.Scene.main(sky=sky, cam=cam).upload(0), ol.OracleScene.main(sky=sky, cam=cam)),
          "terrain": (rt.Scene.terrain(20, seed=5, sky=sky, cam=cam).upload(0), ol.OracleScene.terrain(20, seed=5, sky=sky, cam=cam)),
          "textured": (rt.Scene.textured(sky=sky, cam=cam).upload(0), ol.OracleScene.textured(sky=sky, cam=cam)),
          "spheres": (rt.Scene.spheres(n=300, sky=sky, cam=cam).upload(0), ol.OracleScene.spheres(n=300, sky=sky, cam=cam))}
bad = 0
for name, (g, c) in scenes.items():
    for D, sr, sc, aa, mp, streams in [(2, 1, 1, 3, 1 << 22, 1), (3, 2, 1, 2, 3000, 1), (6, 1, 1, 2, 1 << 22, 2), (1, 3, 1, 3, 5000, 1), (64, 1, 1, 1, 1 << 22, 1),
                                       (2, 1, 2, 2, 4000, 1), (5, 5, 1, 1, 1 << 22, 1), (0, 1, 1, 2, 1 << 22, 1), (4, 1, 0, 1, 1 << 22, 1)]:
        g.set_option(rt.OPT_MAX_PATHS, mp); g.set_option(rt.OPT_STREAMS, streams)
        rect = (7, 5, 131, 97)
        region = rt.Scene.make_region(cam[0], cam[1], rect=rect, tile=(32, 16))
        img, st = g.render(max_depth=D, shadow_rays=sr, scatter_rays=sc, antialiasing=aa, region=region)
        # assemble tiles of the single owner
        full = np.zeros((cam[1], cam[0], 3), np.float32)
        out = np.zeros((rect[3], rect[2], 3), np.float32)
        ref, _, cnt = c.render(max_depth=D, shadow_rays=sr, scatter_rays=sc, antialiasing=aa, rect=rect)
        tiles = img.reshape(-1, 16, 32, 3)
        tx = (rect[2] + 31) // 32
        for t in range(tiles.shape[0]):
            ty, txi = divmod(t, tx)
            blk = tiles[t][: min(16, rect[3] - ty * 16), : min(32, rect[2] - txi * 32)]
            out[ty * 16: ty * 16 + blk.shape[0], txi * 32: txi * 32 + blk.shape[1]] = blk
        d = np.abs(out.astype(np.float64) - ref.astype(np.float64))
        rmse = float(np.sqrt((d ** 2).mean())); frac = float((d > 1e-3).mean())
        ok = rmse <= 2e-3 and frac <= 5e-3 and abs(st["closest_rays"] - cnt["closest_rays"]) <= 2e-3 * max(1, cnt["closest_rays"]) + 2
        bad += not ok
        print(name, (D, sr, sc, aa, mp, streams), "rmse %.2e frac %.2e rays %d/%d %s" % (rmse, frac, st["closest_rays"], cnt["closest_rays"], "ok" if ok else "FAIL"), flush=True)
print("failures:", bad)
