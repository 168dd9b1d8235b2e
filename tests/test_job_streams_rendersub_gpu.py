"""GPU: the rendersub job tests with ONE stream per job (HBHIP_JOB_STREAMS=1; tests/test_job_streams_gpu.py says what the
default is).  Under the default the compositor writes on the frame's own context - decomb's, when decomb made the frame -
behind the readers on the job's other stream: tests/test_rendersub_gpu.py."""
import pytest

from test_rendersub_cpu import registered                            # noqa: F401  (fixture: rendersub + lapsharp registered)
from test_rendersub_gpu import (test_rendersub_inside_a_device_run, test_duplicated_frames_are_composited_once_each)       # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def one_stream(monkeypatch):
    monkeypatch.setenv("HBHIP_JOB_STREAMS", "1")
    yield
