"""GPU: hb_hip_setup_hw_filters puts the drop-ins and the adapters into a job's filter list the way vt_common.c does
for Metal (in place, same ids), and hb_hip_filter_init_failed puts a CPU filter back - and re-brackets the run -
when a drop-in declines its settings.  The pictures equal the all-reference job's (the drop-ins are bit-exact)."""
import numpy as np
import pytest

from handbrake_amd import hbrt, hip, synth
from test_job_swap_cpu import REF, LAP, same, registered      # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu
TFF = 0x0008
F = hbrt.FILTER_ID
NLM = hip.NLMEANS_MEDIUM + ":threads=2"
NLM_P11 = NLM.replace("y-patch-size=7", "y-patch-size=11")      # no kernel for patch 11: the drop-in's init declines
UP, DOWN = "HIP upload adapter", "HIP download adapter"


def run_both(filters, frames, **kw):
    names, out = hbrt.run_job(filters, frames, use_hip=True, **kw)
    _, want = hbrt.run_job(filters, frames, use_hip=False, **kw)
    same(out, want)
    return names


def test_run_of_dropins_is_bracketed_by_adapters(registered):
    frames = synth.stream("interlaced", 320, 180, 5)
    names = run_both([(F["decomb"], "mode=31"), (F["nlmeans"], NLM), (F["lapsharp"], LAP)], frames, flags=TFF)
    assert names == [UP, "Decomb (HIP)", "Denoise (nlmeans, HIP)", "Sharpen (lapsharp, HIP)", DOWN] or \
           (names[0] == UP and names[-1] == DOWN and len(names) == 5 and all("HIP" in n for n in names))


def test_single_dropin_gets_no_adapters(registered):
    frames = synth.stream("progressive", 320, 180, 3)
    names = run_both([(F["lapsharp"], LAP)], frames)
    assert len(names) == 1 and "HIP" in names[0]


def test_declined_filter_in_the_middle_of_a_run_falls_back_to_cpu(registered):
    frames = synth.stream("interlaced", 320, 180, 5)
    names = run_both([(F["decomb"], "mode=7"), (F["nlmeans"], NLM_P11), (F["lapsharp"], LAP), (F["unsharp"], "y-strength=0.25:y-size=7")],
                     frames, flags=TFF)
    # [upload decomb nlm lap unsharp download] -> nlm declines inside the run
    assert names[0] == UP and "Decomb" in names[1] and names[2] == DOWN
    assert names[3] == "Denoise (nlmeans)"
    assert names[4] == UP and "HIP" in names[5] and "HIP" in names[6] and names[7] == DOWN and len(names) == 8


def test_declined_first_filter_of_a_run_undoes_the_upload(registered):
    frames = synth.stream("progressive", 320, 180, 4)
    names = run_both([(F["nlmeans"], NLM_P11), (F["lapsharp"], LAP), (F["unsharp"], "y-strength=0.25:y-size=7")], frames)
    assert names[0] == "Denoise (nlmeans)" and names[1] == UP and names[-1] == DOWN and len(names) == 5


def test_declined_last_filter_leaves_a_single_dropin_without_adapters(registered):
    frames = synth.stream("progressive", 320, 180, 4)
    names = run_both([(F["lapsharp"], LAP), (F["nlmeans"], NLM_P11)], frames)
    # ids order: nlmeans (16) before lapsharp (24): [upload nlm lap download] -> nlm declines first
    assert names[0] == "Denoise (nlmeans)" and len(names) == 2 and "HIP" in names[1]


def test_comb_detect_then_selective_decomb_device_resident(registered):
    frames = synth.stream("interlaced", 320, 180, 6)
    names = run_both([(F["comb_detect"], ""), (F["decomb"], "mode=39"), (F["denoise"], "y-spatial=3")], frames, flags=TFF)
    assert names[0] == UP and names[-1] == DOWN and len(names) == 5
