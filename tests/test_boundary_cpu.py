"""CPU: the drop-in boundary against the reference's REAL headers.

The HIP filter objects (handbrake_amd/libhb/*_hip.c, hip_common.c, hbhip_registry.c) are compiled with
-DHBHIP_IN_LIBHB, i.e. against /root/reference/libhb/handbrake/handbrake.h + internal.h instead of our own
re-declaration (include/hbhip_libhb.h): every libhb type, field, enum and function they use must exist there with a
compatible type.  libav* / jansson are not in this image; tests/libhb_stubs/ stands in for the handful of their
names libhb's headers mention (opaque types - only names and kinds matter for a syntax + type check, nothing links).
The three enumerators INTEGRATION.md adds to libhb come in as -D, exactly as the patch there spells them.
Skipped where /root/reference does not exist (the GPU box)."""
import glob
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/libhb"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference tree")

# what INTEGRATION.md section 1 adds to libhb: a storage type next to COREMEDIA (internal.h:152-153) and two filter ids
# behind the last one (common.h:1729-1778)
PATCH_DEFINES = ["-DHBHIP_DEVICE=3", "-DHB_FILTER_HIP_UPLOAD=98", "-DHB_FILTER_HIP_DOWNLOAD=99"]
FLAGS = ["-fsyntax-only", "-std=gnu99", "-Wall", "-Werror=implicit-function-declaration", "-Werror=incompatible-pointer-types",
         "-Werror=int-conversion", "-D__LIBHB__", "-DHBHIP_IN_LIBHB", f"-I{ROOT}/tests/libhb_stubs", f"-I{REF}",
         f"-I{ROOT}/include", f"-I{ROOT}/handbrake_amd/libhb"] + PATCH_DEFINES
SOURCES = sorted(glob.glob(os.path.join(ROOT, "handbrake_amd", "libhb", "*_hip.c"))) + \
    [os.path.join(ROOT, "handbrake_amd", "libhb", f) for f in ("hip_common.c", "hbhip_registry.c")]


@pytest.mark.parametrize("src", SOURCES, ids=[os.path.basename(s) for s in SOURCES])
def test_drop_in_compiles_against_the_real_libhb_headers(src):
    r = subprocess.run(["gcc"] + FLAGS + [src], capture_output=True, text=True)
    errors = [ln for ln in r.stderr.splitlines() if "error" in ln]
    assert r.returncode == 0, "\n".join(errors[:20])


OFFSETS = r"""
#include <stddef.h>
#include <stdio.h>
%s
#define F(field) printf(#field " %%zu\n", offsetof(hb_filter_object_t, field))
int main(void)
{
    F(id); F(enforce_order); F(skip); F(aliased); F(name); F(short_name); F(settings); F(init); F(init_thread); F(post_init);
    F(work); F(work_thread); F(close); F(info); F(settings_template); F(fifo_in); F(fifo_out); F(private_data); F(thread);
    F(done); F(status); F(chapter_time); F(chapter_val); F(sub_filter);
    printf("sizeof %%zu\n", sizeof(hb_filter_object_t));
    return 0;
}
"""


def test_filter_object_layout_equals_the_reference(tmp_path):
    """struct hb_filter_object_s (common.h:1670-1711) field for field: the same offsets in the reference's header and
    in include/hbhip_libhb.h - the object files of the two worlds could be linked against each other."""
    outs = []
    for tag, inc, flags in (("real", '#include "handbrake/handbrake.h"', ["-D__LIBHB__", f"-I{ROOT}/tests/libhb_stubs", f"-I{REF}"]),
                            ("ours", '#include "hbhip_libhb.h"', [f"-I{ROOT}/include"])):
        c = tmp_path / f"off_{tag}.c"
        c.write_text(OFFSETS % inc)
        exe = tmp_path / f"off_{tag}"
        subprocess.check_call(["gcc", "-std=gnu99"] + flags + [str(c), "-o", str(exe)])
        outs.append(subprocess.check_output([str(exe)], text=True))
    assert outs[0] == outs[1], f"reference:\n{outs[0]}\nours:\n{outs[1]}"
