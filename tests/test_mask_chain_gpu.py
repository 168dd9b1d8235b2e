"""GPU: the EEDI2 mask chain (csrc/eedi2_engine.h: MaskChain) can never abort the process.

The lower mask tiles of all fields of a batch run as ONE launch whose tiles wait, through flags in device memory, for
the previous field's tiles.  A wait is bounded; when it runs out the tile raises an error word and goes on, the launch
ends, and a one-workgroup repair pass queued behind it recomputes the batch's masks field by field (no trap, no host
round trip).  Here:
  * the repair path itself, forced by giving every wait a bound of one poll (hbhip_debug_mask_chain(1)): the frames
    must still be the oracle's, at 8 and 10 bits, through the plugin and through a 16-link chain at 1080p;
  * the chain under contention: two chains on two contexts (HIP streams) and a third stream that keeps every CU busy
    with a filler kernel, twenty repetitions at 1080p, every output identical to the first repetition's;
  * no s_trap instruction in any code object of the product library (tools/no_trap_check.sh)."""
import os
import subprocess

import numpy as np
import pytest

from handbrake_amd import hbrt, hip, synth
import oracle_stream as os_

pytestmark = pytest.mark.gpu
TFF = 0x0008
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def chain_debug(limit):
    import ctypes as C
    fn = hip.lib().hbhip_debug_mask_chain
    fn.restype = C.c_uint
    fn.argtypes = [C.c_int]
    return fn(limit)


@pytest.fixture()
def one_poll(built):
    before = chain_debug(1)
    yield before
    chain_debug(0)


@pytest.mark.parametrize("depth", [8, 10])
def test_repair_path_through_the_plugin_gives_the_oracles_frames(one_poll, depth):
    w, h, n = 322, 184, 7
    frames = synth.stream("interlaced", w, h, n, depth=depth)
    got = hbrt.run_stream(hip.filters(), [("hb_filter_decomb_hip", "mode=31")], frames, flags=TFF,
                          pix_fmt=hbrt.PIX_FMT_FOR_DEPTH[depth])
    res = os_.decomb_eedi2_stream(frames, dict(mode=31, depth=depth), flags=TFF)
    assert len(got) == len(res) == 2 * n
    for t in range(len(res)):
        for c in range(3):
            np.testing.assert_array_equal(got[t].planes[c], res[t]["planes"][c], err_msg=f"frame {t} plane {c}")
    assert chain_debug(-1) > one_poll, "the repair pass never ran although every wait was given one poll"


def _run_chain_object(ctx, w, h, depth, dev_in, n, torch):
    dt = torch.uint8 if depth == 8 else torch.uint16
    cap = 2 * n + 2
    outs = [[torch.zeros((h, w), dtype=dt, device="cuda"), torch.zeros((h // 2, w // 2), dtype=dt, device="cuda"),
             torch.zeros((h // 2, w // 2), dtype=dt, device="cuda")] for _ in range(cap)]
    torch.cuda.synchronize()
    dec = hip.DecombDevice(ctx, w, h, mode=31, postproc=1, depth=depth)
    stage = hip.DeviceFilter(ctx, dec.h)
    dec.h = None
    chain = hip.Chain(ctx, [stage])
    try:
        arr_in = (hip.DevFrame * n)(*[hip.dev_frame(f) for f in dev_in])
        arr_out = (hip.DevFrame * cap)(*[hip.dev_frame(o) for o in outs])
        k = chain.process_dev(arr_in, arr_out, tag0=0, flags=[TFF] * n, combed=[2] * n)
        chain.sync()
        got = [[p.clone() for p in outs[i]] for i in range(k)]
        k = chain.flush_dev(arr_out)
        chain.sync()
        got += [[p.clone() for p in outs[i]] for i in range(k)]
    finally:
        chain.close()
    return got


@pytest.mark.parametrize("depth", [8, 10])
def test_repair_path_of_a_16_link_chain_at_1080p(built, depth):
    """12 frames = 24 fields in one batch (parts of 16 and 8 links): first with the default bound, then with every
    wait given one poll - the same frames, and the first four of them pinned by the oracle."""
    import torch
    w, h, n = 1920, 1080, 12
    frames = synth.stream("interlaced", w, h, n, depth=depth)
    ctx = hip.Ctx(0)
    try:
        dev_in = [[torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in f] for f in frames]
        normal = _run_chain_object(ctx, w, h, depth, dev_in, n, torch)
        before = chain_debug(1)
        try:
            forced = _run_chain_object(ctx, w, h, depth, dev_in, n, torch)
        finally:
            chain_debug(0)
        # (the count is read by the engines' launch() / destructor: the chain object above is closed by now)
        assert chain_debug(-1) > before
        assert len(normal) == len(forced) == 2 * n
        for i in range(2 * n):
            for c in range(3):
                assert torch.equal(normal[i][c], forced[i][c]), f"frame {i} plane {c}: repair path differs"
        want = os_.decomb_eedi2_stream(frames[:3], dict(mode=31, postproc=1, depth=depth), flags=TFF)
        for i in range(4):
            for c in range(3):
                np.testing.assert_array_equal(forced[i][c].cpu().numpy(), want[i]["planes"][c], err_msg=f"frame {i} plane {c}")
    finally:
        ctx.close()


def test_two_chains_and_a_cu_filling_stream_twenty_times(built):
    """Two mask chains in flight on two contexts while a third stream keeps the CUs full: whatever the dispatcher does
    with three queues, every repetition's frames equal the first's (and a wait that ran out would have been repaired,
    not trapped).  The number of repaired launches is printed; it is expected to stay where it was."""
    import torch
    w, h, n, reps = 1920, 1080, 8, 20
    fa = synth.stream("interlaced", w, h, n, cfg=3)
    fb = synth.stream("interlaced", w, h, n, cfg=5)
    ctxs = [hip.Ctx(0), hip.Ctx(0)]
    before = chain_debug(-1)
    cap = 2 * n + 2
    try:
        dev = [[[torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in f] for f in fr] for fr in (fa, fb)]
        outs = [[[torch.zeros((h, w), dtype=torch.uint8, device="cuda"), torch.zeros((h // 2, w // 2), dtype=torch.uint8, device="cuda"),
                  torch.zeros((h // 2, w // 2), dtype=torch.uint8, device="cuda")] for _ in range(cap)] for _ in range(2)]
        filler = torch.ones(64 * 1024 * 1024, dtype=torch.float32, device="cuda")          # 256 MB: 65 536 workgroups per pass
        side = torch.cuda.Stream()
        torch.cuda.synchronize()
        chains = []
        for c in ctxs:
            dec = hip.DecombDevice(c, w, h, mode=31)
            st = hip.DeviceFilter(c, dec.h)
            dec.h = None
            chains.append(hip.Chain(c, [st]))
        first = None
        try:
            for rep in range(reps):
                with torch.cuda.stream(side):
                    for _ in range(24):
                        filler.mul_(1.0000001)
                ks = []
                for i, ch in enumerate(chains):                            # both batches queued before either is waited for
                    arr_in = (hip.DevFrame * n)(*[hip.dev_frame(f) for f in dev[i]])
                    arr_out = (hip.DevFrame * cap)(*[hip.dev_frame(o) for o in outs[i]])
                    ks.append(ch.process_dev(arr_in, arr_out, tag0=rep * n, flags=[TFF] * n, combed=[2] * n))
                for ch in chains:
                    ch.sync()
                got = [[[p.clone() for p in outs[i][j]] for j in range(ks[i])] for i in range(2)]
                # the stream goes on: the next repetition's first field reads this one's last mask, so every repetition
                # feeds the same frames into a DIFFERENT state - compare like with like: flush and reopen per repetition
                for i, ch in enumerate(chains):
                    arr_out = (hip.DevFrame * cap)(*[hip.dev_frame(o) for o in outs[i]])
                    k = ch.flush_dev(arr_out)
                    ch.sync()
                    got[i] += [[p.clone() for p in outs[i][j]] for j in range(k)]
                    ch.close()
                chains = []
                for c in ctxs:
                    dec = hip.DecombDevice(c, w, h, mode=31)
                    st = hip.DeviceFilter(c, dec.h)
                    dec.h = None
                    chains.append(hip.Chain(c, [st]))
                side.synchronize()
                if first is None:
                    first = got
                    assert len(first[0]) == len(first[1]) == 2 * n
                    continue
                for i in range(2):
                    assert len(got[i]) == len(first[i])
                    for j in range(len(first[i])):
                        for c in range(3):
                            assert torch.equal(got[i][j][c], first[i][j][c]), f"repetition {rep} chain {i} frame {j} plane {c}"
        finally:
            for ch in chains:
                ch.close()
        want = os_.decomb_eedi2_stream(fa[:2], dict(mode=31), flags=TFF)
        for j in range(2):
            for c in range(3):
                np.testing.assert_array_equal(first[0][j][c].cpu().numpy(), want[j]["planes"][c])
        print("mask-chain launches repaired during the stress run:", chain_debug(-1) - before)
    finally:
        for c in ctxs:
            c.close()


def test_product_device_code_has_no_trap_instruction(built):
    tool = os.path.join(ROOT, "tools", "no_trap_check.sh")
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("ROCm binutils not present")
    r = subprocess.run(["bash", tool], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1] == "0", r.stdout + r.stderr
