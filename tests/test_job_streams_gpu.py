"""GPU: a second HIP stream inside a job (the default since round 6) against HBHIP_JOB_STREAMS=1.  The deinterlacing side of a job's list (comb detect, decomb,
yadif, bwdif) gets a context of its own beside the job's (hbhip_host_ctx_for_role), the way bench.py's device-resident line
runs the chain (--stage-streams 2); frames then cross contexts, ordered by hbhip_frame_use_on behind their producer and
going idle behind their last reader's stream.  Every job-level test runs under the default elsewhere; here they run again with
ONE stream per job (HBHIP_JOB_STREAMS=1): the pictures must not depend on it."""
import pytest

from handbrake_amd import hbrt, hip, synth
from test_job_swap_cpu import registered, same                      # noqa: F401  (fixtures)
from test_job_swap_gpu import with_vfr                               # noqa: F401
from test_job_swap_gpu import (test_run_of_dropins_is_bracketed_by_adapters, test_declined_filter_in_the_middle_of_a_run_falls_back_to_cpu,          # noqa: F401
                               test_comb_detect_then_selective_decomb_device_resident, test_vfr_stays_inside_the_device_run,
                               test_declined_dropin_behind_vfr_closes_the_run_after_it, test_configs3_job_with_vfr_1080i_to_2160p,
                               test_concurrent_jobs_on_one_gpu_run_on_streams_of_their_own)
from test_formats_gpu import test_device_resident_job_on_yuv422p10    # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def one_stream(monkeypatch):
    monkeypatch.setenv("HBHIP_JOB_STREAMS", "1")
    yield


def test_decomb_really_sits_on_a_context_of_its_own(registered, monkeypatch):
    """the job's own context and the deinterlacing side's differ by default, and are one under HBHIP_JOB_STREAMS=1"""
    import ctypes as C
    flt = hip.filters()
    flt.hbhip_host_ctx_for_role.restype = C.c_void_p
    flt.hbhip_host_ctx_for_role.argtypes = [C.c_void_p, C.c_int]
    F = hbrt.FILTER_ID
    with hbrt.Job([(F["decomb"], "mode=7"), (F["lapsharp"], "y-strength=0.2:y-kernel=isolap")], 320, 180, use_hip=True) as job:
        init = (C.c_void_p * 32)()
        init[0] = job.job_ptr()
        a = flt.hbhip_host_ctx_for_role(init, 0)
        assert a is not None and flt.hbhip_host_ctx_for_role(init, 1) == a            # one stream (this module's setting)
        monkeypatch.delenv("HBHIP_JOB_STREAMS")
        b = flt.hbhip_host_ctx_for_role(init, 1)
        assert b is not None and b != a                                               # the default: two


@pytest.mark.parametrize("w,h,n", [(1920, 1080, 96), (320, 180, 400)])
def test_two_streams_equal_one_stream_over_a_long_run(with_vfr, monkeypatch, w, h, n):
    """A long run at full speed through [comb detect, decomb 63, vfr (duplicates), nlmeans, lapsharp], one thread per filter:
    frames cross from the job's context to decomb's and back while both streams are busy, pool frames are recycled many
    times over (hbhip_frame_use_on's idle marks) and the upload adapter has copies in flight - the pictures of the default
    (two streams) must equal those of one stream, frame for frame."""
    base = synth.stream("interlaced", w, h, 8, cfg=3)
    frames = [base[i % 8] for i in range(n)]
    F = hbrt.FILTER_ID
    lst = [(F["comb_detect"], ""), (F["decomb"], "mode=63"), (11, "mode=1:rate=60000/1001"),
           (F["nlmeans"], hip.NLMEANS_MEDIUM + ":threads=2"), (F["lapsharp"], "y-strength=0.2:y-kernel=isolap")]
    _, one = hbrt.run_job(lst, frames, flags=0x0008, use_hip=True)              # this module's setting: one stream
    monkeypatch.delenv("HBHIP_JOB_STREAMS")
    _, two = hbrt.run_job(lst, frames, flags=0x0008, use_hip=True)
    assert len(one) == len(two) >= n
    same(two, one)


def test_adopted_frames_equal_copied_frames(with_vfr, monkeypatch):
    """Inside a device-resident run decomb and NLMeans take the frames they are pushed as their input pictures and hand their
    result pictures on as frames (hbhip_filter_use_frames / push_frame / pull_frame); HBHIP_ZERO_COPY=0 makes them copy into
    and out of pictures of their own, as they did until round 6.  Same pictures either way - with vfr's duplicates in the
    list (shared frames are copied, not adopted) and with two streams per job (adopted frames cross contexts)."""
    monkeypatch.delenv("HBHIP_JOB_STREAMS")
    w, h, n = 640, 360, 40
    base = synth.stream("interlaced", w, h, 8, cfg=3)
    frames = [base[i % 8] for i in range(n)]
    F = hbrt.FILTER_ID
    lst = [(F["comb_detect"], ""), (F["decomb"], "mode=31"), (11, "mode=1:rate=90000/1001"),
           (F["nlmeans"], hip.NLMEANS_MEDIUM + ":threads=2"), (F["lapsharp"], "y-strength=0.2:y-kernel=isolap")]
    _, adopted = hbrt.run_job(lst, frames, flags=0x0008, use_hip=True)
    monkeypatch.setenv("HBHIP_ZERO_COPY", "0")
    _, copied = hbrt.run_job(lst, frames, flags=0x0008, use_hip=True)
    assert len(adopted) == len(copied) > 2 * n          # 90000/1001 = three out of every bobbed pair: duplicates
    same(adopted, copied)
