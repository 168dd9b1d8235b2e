import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from handbrake_amd import hbrt, hip, synth
import oracle_lib as ol
hbrt_filters = hip.filters()
src, dst = (1, 1, 1, 1), (9, 18, 9, 1)
w, h = 638, 362
frames = synth.stream("progressive", w, h, 1)
hbrt.set_source_color(*src)
got = hbrt.run_stream(hbrt_filters, [("hb_filter_colorspace_hip", "primaries=bt2020:transfer=arib-std-b67:matrix=bt2020nc")], frames)
hbrt.set_source_color()
want = ol.orc_colorspace_frame(frames[0], ol.colorspace_params(src, dst))
for c in range(3):
    bad = np.argwhere(got[0].planes[c] != want[c])
    for (y, x) in bad[:8]:
        print("plane", c, "y", y, "x", x, "got", got[0].planes[c][y, x], "want", want[c][y, x])
        if c:
            print(" luma in", frames[0][0][2*y-1:2*y+3, 2*x-1:2*x+2].tolist())
            print(" u in", frames[0][1][y-1:y+2, x-1:x+2].tolist(), "v in", frames[0][2][y-1:y+2, x-1:x+2].tolist())
