"""GPU: vfr's frame-difference metric (hb_motion_metric_hip / hbhip_motion_metric_*) against
oracle/motion_metric_oracle.c, which is pinned to the reference's own object.  Exact equality: the
sums are integers and the final division is the reference's float expression."""
import numpy as np
import pytest

from handbrake_amd import hbrt, hip, synth
import oracle_lib as ol

pytestmark = pytest.mark.gpu
SIZES = [(128, 72), (638, 362), (1280, 720), (1920, 1080), (1918, 1078), (720, 1088), (3840, 2160)]


@pytest.mark.parametrize("depth", [8, 10, 12])
@pytest.mark.parametrize("w,h", SIZES)
@pytest.mark.parametrize("model", ["progressive", "random"])
def test_matches_oracle(built, model, w, h, depth):
    fr = synth.stream(model, w, h, 2, depth=depth)
    for a, b in ((fr[0][0], fr[1][0]), (fr[1][0], fr[1][0])):
        got = hbrt.motion_metric_run(hip.filters(), "hb_motion_metric_hip", a, b, pix_fmt=hbrt.PIX_FMT_FOR_DEPTH[depth])
        assert got == ol.orc_motion_metric(a, b, depth), (model, w, h, depth)


def test_extreme_blocks_wrap_like_the_reference(built):
    a = np.zeros((64, 64), np.uint8)
    b = np.full((64, 64), 255, np.uint8)
    assert hbrt.motion_metric_run(hip.filters(), "hb_motion_metric_hip", a, b) == ol.orc_motion_metric(a, b)


def test_device_resident_planes(built):
    import torch
    w, h = 1920, 1080
    fr = synth.stream("interlaced", w, h, 2)
    L = hip.lib()
    C = hip.C
    lut = (C.c_uint * 256)()
    ol.oracle().orc_motion_gamma_lut(lut, 8)
    L.hbhip_motion_metric_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    L.hbhip_motion_metric_run_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_float)]
    L.hbhip_motion_metric_destroy.argtypes = [C.c_void_p]
    ctx = hip.Ctx(0)
    m = C.c_void_p()
    try:
        assert L.hbhip_motion_metric_create(ctx.h, w, h, 8, lut, 256, C.byref(m)) == 0
        ta = torch.from_numpy(np.ascontiguousarray(fr[0][0])).to("cuda:0")
        tb = torch.from_numpy(np.ascontiguousarray(fr[1][0])).to("cuda:0")
        torch.cuda.synchronize()
        out = C.c_float()
        assert L.hbhip_motion_metric_run_dev(m, ta.data_ptr(), ta.stride(0), tb.data_ptr(), tb.stride(0), C.byref(out)) == 0
        assert out.value == ol.orc_motion_metric(fr[0][0], fr[1][0])
    finally:
        if m:
            L.hbhip_motion_metric_destroy(m)
        ctx.close()
