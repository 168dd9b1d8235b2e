"""GPU: frames stay in HBM between adjacent HIP filters (hb_filter_hip_upload ... hb_filter_hip_download
around the chain) and the output is byte-identical to the per-filter host path."""
import numpy as np
import pytest

from handbrake_amd import hbrt, hip, synth
import golden_cases as gc

pytestmark = pytest.mark.gpu
TFF = 0x0008
UP, DOWN = ("hb_filter_hip_upload", ""), ("hb_filter_hip_download", "")

CHAINS = {
    "chain4": [("hb_filter_decomb_hip", "mode=31"), ("hb_filter_nlmeans_hip", hip.NLMEANS_MEDIUM),
               ("hb_filter_crop_scale_hip", "width=1280:height=720"),
               ("hb_filter_lapsharp_hip", "y-strength=0.2:y-kernel=isolap:cb-strength=0.2:cb-kernel=isolap")],
    "comb+decomb": [("hb_filter_comb_detect_hip", gc.COMB_DEFAULT), ("hb_filter_decomb_hip", "mode=55")],
    "denoise+sharpen+gray+rotate": [("hb_filter_denoise_hip", ""), ("hb_filter_chroma_smooth_hip", "cb-strength=0.6:cb-size=7"),
                                    ("hb_filter_unsharp_hip", "y-strength=0.25:y-size=7"),
                                    ("hb_filter_rotate_hip", "angle=90:hflip=1"),
                                    ("hb_filter_grayscale_hip", "cb=0:cr=0:size=1:high=0")],
}


@pytest.mark.parametrize("name", sorted(CHAINS))
@pytest.mark.parametrize("w,h", [(640, 360), (638, 360)])
def test_device_resident_chain_equals_host_chain(built, name, w, h):
    frames = synth.stream("interlaced", w, h, 6)
    chain = CHAINS[name]
    host = hbrt.run_stream(hip.filters(), chain, frames, flags=TFF)
    dev = hbrt.run_stream(hip.filters(), [UP] + chain + [DOWN], frames, flags=TFF)
    assert len(dev) == len(host) > 0
    for t in range(len(host)):
        assert (dev[t].start, dev[t].stop, dev[t].combed) == (host[t].start, host[t].stop, host[t].combed)
        for c in range(3):
            np.testing.assert_array_equal(dev[t].planes[c], host[t].planes[c], err_msg=f"{name} frame {t} plane {c}")
