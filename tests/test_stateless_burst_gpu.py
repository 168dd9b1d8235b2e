"""GPU: the stateless drop-ins inside a device-resident run gather a burst of frames per launch (hbhip_host_simple_work,
HBHIP_STATELESS_BATCH, default 8).  The pictures, their order and their timestamps must not depend on the burst size - also
when the stream's length is not a multiple of it (the EOF flushes what was gathered), when a duplicated frame sits in a
burst twice (vfr), and for the filters whose state runs from frame to frame (hqdn3d).  The job-level tests elsewhere run
under the default; here some run again with a launch per frame."""
import pytest

from handbrake_amd import hbrt, hip, synth
from test_job_swap_cpu import registered, same                      # noqa: F401  (fixtures)
from test_job_swap_gpu import with_vfr                               # noqa: F401
from test_job_swap_gpu import (test_run_of_dropins_is_bracketed_by_adapters, test_vfr_stays_inside_the_device_run,          # noqa: F401
                               test_configs3_job_with_vfr_1080i_to_2160p)

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def a_launch_per_frame(monkeypatch):
    monkeypatch.setenv("HBHIP_STATELESS_BATCH", "1")
    yield


@pytest.mark.parametrize("n", [5, 8, 21])
@pytest.mark.parametrize("burst", ["3", "8", "16"])
def test_burst_size_does_not_show(with_vfr, monkeypatch, n, burst):
    """[decomb 7, vfr with CFR duplicates, hqdn3d, crop_scale down, lapsharp, unsharp] as one device-resident run: a launch
    per frame against bursts of 3 / 8 / 16, stream lengths on either side of a burst"""
    base = synth.stream("interlaced", 640, 360, 6, cfg=3)
    frames = [base[i % 6] for i in range(n)]
    F = hbrt.FILTER_ID
    lst = [(F["decomb"], "mode=7"), (11, "mode=1:rate=60000/1001"), (F["denoise"], "y-spatial=3"),
           (F["crop_scale"], "width=320:height=180"), (F["lapsharp"], "y-strength=0.2:y-kernel=isolap"),
           (F["unsharp"], "y-strength=0.25:y-size=7")]
    _, one = hbrt.run_job(lst, frames, flags=0x0008, use_hip=True)              # this module's setting: a launch per frame
    monkeypatch.setenv("HBHIP_STATELESS_BATCH", burst)
    _, many = hbrt.run_job(lst, frames, flags=0x0008, use_hip=True)
    assert len(one) == len(many) >= n
    same(many, one)
