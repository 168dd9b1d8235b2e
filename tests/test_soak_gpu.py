"""GPU: the drop-ins over many jobs and over a long stream - device memory must reach a steady state.  A libhb process
lives for a whole queue of encodes: contexts and their pools are kept for the life of the process (hbhip_registry.c), so
anything a job allocates beyond what its filters free at close() would add up, and a pool that grew with the stream's
length would end an encode hours in."""
import numpy as np
import pytest
import torch

from handbrake_amd import hbrt, hip, synth
from test_job_swap_cpu import REF, LAP, same, registered      # noqa: F401  (fixture)

import os

# device-wide free memory is what HIP can report (no per-process figure in this container): the tests need the GPU to
# themselves - they run in the serial suite (what the driver runs) and are skipped under pytest-xdist
pytestmark = [pytest.mark.gpu, pytest.mark.skipif("PYTEST_XDIST_WORKER" in os.environ,
                                                    reason="measures device-wide free memory: serial runs only")]
TFF = 0x0008
F = hbrt.FILTER_ID
NLM = hip.NLMEANS_MEDIUM + ":threads=2"
MB = 1 << 20


def free_mb():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] / MB


def test_a_queue_of_jobs_does_not_grow_device_memory(registered):
    frames = synth.stream("interlaced", 320, 180, 6)
    lst = [(F["comb_detect"], ""), (F["decomb"], "mode=63"), (F["nlmeans"], NLM), (F["denoise"], "y-spatial=3"),
           (F["unsharp"], "y-strength=0.25:y-size=7"), (F["lapsharp"], LAP)]
    first = None
    marks = []
    for i in range(14):
        _, out = hbrt.run_job(lst, frames, flags=TFF, use_hip=True)
        if first is None:
            first = out
        else:
            same(out, first)
        marks.append(free_mb())
    # the first jobs fill the context's pools; from then on a job gives back what it took
    assert marks[3] - marks[-1] < 48, f"free device memory fell from {marks[3]:.0f} to {marks[-1]:.0f} MB over 10 jobs: {marks}"


@pytest.mark.parametrize("w,h,n,slack", [(320, 180, 600, 32), (1920, 1080, 240, 256)])
def test_a_long_stream_reaches_a_steady_state(registered, w, h, n, slack):
    """n frames through [decomb 31, nlmeans, lapsharp] as one device-resident run: the frame pool is capped (48 frames per
    shape), EEDI2's slots and the filters' rings are fixed - free memory a quarter of the way in and at the end differ by
    pool noise only (1080p: a pool frame is 3 MB, EEDI2's slab is allocated at init)"""
    base = synth.stream("interlaced", w, h, 8)
    lst = [(F["decomb"], "mode=31"), (F["nlmeans"], NLM), (F["lapsharp"], LAP)]
    n_out = 0
    with hbrt.Job(lst, w, h, use_hip=True) as job:
        mark = None
        for i in range(n):
            job.push(base[i % 8], start=i * 3003, stop=(i + 1) * 3003, flags=TFF)
            n_out += len(job.drain())
            if i == n // 4:
                mark = free_mb()
        end = free_mb()
        job.push_eof()
        n_out += len(job.drain())
    assert n_out == 2 * n
    assert mark - end < slack, f"free device memory fell from {mark:.0f} to {end:.0f} MB between frame {n // 4} and frame {n}"
