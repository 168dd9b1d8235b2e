"""GPU: the measurement helpers bench.py leans on give sane numbers on this box."""
import ctypes as C

import pytest

from handbrake_amd import hip

pytestmark = pytest.mark.gpu


def test_copy_bandwidth_of_the_library_is_in_the_hbm_range(built):
    """hbhip_ctx_copy_bandwidth: a float4 copy over two 512 MB buffers (past the 256 MB Infinity Cache): between 2 and
    8 TB/s on an MI355X (the guide measures 6.29 with the same kind of kernel; the nominal peak is 8)."""
    ctx = hip.Ctx(0)
    try:
        fn = hip.lib().hbhip_ctx_copy_bandwidth
        fn.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_double)]
        fn.restype = C.c_int
        out = C.c_double()
        assert fn(ctx.h, 512 << 20, 3, C.byref(out)) == 0
        assert 2000.0 < out.value < 8000.0, out.value
        assert fn(ctx.h, 16, 3, C.byref(out)) != 0              # too small to mean anything: refused
    finally:
        ctx.close()
