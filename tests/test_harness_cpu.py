"""CPU: the stand-in harness's source (hbh_chain_feed: C threads fill the hb_buffer_t's, frames pushed in order - what the
PCIe-inclusive bench pass feeds its pipeline with) delivers exactly what frame-by-frame pushes deliver."""
import numpy as np
import pytest

from handbrake_amd import hbrt, synth
import oracle_lib as ol

LAP = "y-strength=0.2:y-kernel=isolap:cb-strength=0.2:cb-kernel=isolap"


@pytest.mark.parametrize("threaded", [False, True])
@pytest.mark.parametrize("threads", [1, 3])
def test_feed_equals_push(built, threaded, threads):
    if ol.ref() is None:
        pytest.skip("oracle/_ref not built (no /root/reference)")
    uniq = synth.stream("progressive", 322, 182, 5)                      # 322: rows that are no multiple of the 64-byte stride
    n, first = 23, 7
    chain = [("hb_filter_lapsharp", LAP), ("hb_filter_unsharp", "y-strength=0.25:y-size=7")]
    hbrt.set_threaded(threaded)
    try:
        with hbrt.Chain(ol.ref(), chain, 322, 182) as ch:
            ch.feed(uniq, first, n, duration=3003, flags=0x10, threads=threads)
            ch.push_eof()
            got = ch.drain()
        with hbrt.Chain(ol.ref(), chain, 322, 182) as ch:
            want = []
            for i in range(first, first + n):
                ch.push(uniq[i % 5], start=i * 3003, stop=(i + 1) * 3003, flags=0x10)
                want += ch.drain()
            ch.push_eof()
            want += ch.drain()
    finally:
        hbrt.set_threaded(False)
    assert len(got) == len(want) == n
    for g, w in zip(got, want):
        assert (g.start, g.stop, g.flags) == (w.start, w.stop, w.flags)
        for c in range(3):
            np.testing.assert_array_equal(g.planes[c], w.planes[c])
