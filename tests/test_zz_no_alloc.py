"""rtamd_trace_closest_static (the entry point that allocates nothing) against rtamd_trace_closest_device.

Both run the same scheduler kernel; the only difference is where the kernel's queue-head word lives (the library's
static device array instead of a heap allocation of the context).  Bar: every output array bit-identical, on two calls
in a row, the second with a smaller input, plus the size limit's error path.
"""
import ctypes as C

import numpy as np
import pytest

from test_gpu_trace import check_closest, random_rays

pytestmark = pytest.mark.gpu


def _device_batch(torch, rays):
    d = torch.from_numpy(rays).cuda()
    n = rays.shape[1]
    outs = [torch.full((n,), 0x7FC0FFEE, dtype=torch.int32, device="cuda") for _ in range(5)]   # poison: stale results would show
    return d, outs


def _as_host(outs):
    names = ("obj", "prim", "t", "u", "v")
    return {k: o.cpu().numpy().view(np.uint32 if k in ("obj", "prim") else np.float32) for k, o in zip(names, outs)}


def test_static_entry_point_matches_device_entry_point(rt, ol, sky):
    import torch
    cam = (160, 120)
    gpu = rt.Scene.main(sky=sky, cam=cam).upload(0)
    cpu = ol.OracleScene.main(sky=sky, cam=cam)
    for n, seed in ((50_000, 41), (7_001, 43)):   # second call: smaller input, the ring's next head word
        rays = random_rays(n, seed=seed)
        d, ref_out = _device_batch(torch, rays)
        _, got_out = _device_batch(torch, rays)
        torch.cuda.synchronize()
        gpu.trace_closest_device([d[i] for i in range(7)], ref_out, n)
        gpu.trace_closest_static([d[i] for i in range(7)], got_out, n)
        torch.cuda.synchronize()
        ref, got = _as_host(ref_out), _as_host(got_out)
        for k in ref:
            assert np.array_equal(got[k].view(np.uint32), ref[k].view(np.uint32)), (n, k)
        check_closest(rt, got, cpu.trace_closest(rays))
    # on a caller stream: asynchronous, still the same bits
    rays = random_rays(3_000, seed=47)
    d, ref_out = _device_batch(torch, rays)
    _, got_out = _device_batch(torch, rays)
    torch.cuda.synchronize()
    gpu.trace_closest_device([d[i] for i in range(7)], ref_out, rays.shape[1])
    s = torch.cuda.Stream()
    gpu.trace_closest_static([d[i] for i in range(7)], got_out, rays.shape[1], stream=C.c_void_p(s.cuda_stream))
    s.synchronize()
    ref, got = _as_host(ref_out), _as_host(got_out)
    for k in ref:
        assert np.array_equal(got[k].view(np.uint32), ref[k].view(np.uint32)), k
    # beyond the stated maximum: an error code, nothing launched
    with pytest.raises(rt.RtamdError, match="RTAMD_STATIC_MAX_RAYS"):
        gpu.trace_closest_static([d[i] for i in range(7)], got_out, 1 << 31)
    gpu.trace_closest_static([d[i] for i in range(7)], got_out, 0)   # empty batch: no-op
