"""The call-tape machinery (optispeech_amd/fastcall.py -> _ospfast, optispeech_amd/tape.py) on the CPU: record, patch, replay.

``osp_tape_selftest`` / ``osp_tape_selftest_table`` are host-only entry points of the library (no device work), so the recording /
patching / replay logic is testable without a GPU; the `-m gpu` suite checks the taped regions of the training step against their
eager execution bit for bit (tests/test_gpu_tape.py)."""
import numpy as np
import pytest

from optispeech_amd import _lib


@pytest.fixture(scope="module")
def fast():
    f = _lib.lib()._fast
    if f is None or not hasattr(f, "tape_begin"):
        pytest.fail("_ospfast with tape support is not built (python -m optispeech_amd.build)")
    f.set_guard(False)                       # host addresses go where a device pointer is declared (the selftest entry points)
    yield f
    f.tape_abort()
    f.set_guard(True)


def _addr(a, i=0):
    return a.ctypes.data + 8 * i


def test_record_and_replay_follow_the_inputs_and_the_stream(fast):
    L = _lib.lib()
    idx = L._fidx["osp_tape_selftest"]
    a = np.zeros(4, dtype=np.int64)           # the region's "input": counters live inside it
    b = np.zeros(4, dtype=np.int64)           # a buffer that is NOT an input: its address stays baked in
    fast.tape_begin([_addr(a)], [a.nbytes], 7)
    assert fast.tape_recording()
    assert fast.call(idx, 7, _addr(a, 1), 10) == 0            # on the recording's current stream (7) -> follows the replay stream
    assert fast.call(idx, 3, _addr(a, 2), 100) == 0           # on another stream (3) -> keeps it
    assert fast.call(idx, 7, _addr(b, 0), 1000) == 0          # outside every input -> address kept
    tape = fast.tape_end()
    assert not fast.tape_recording()
    # the recording run executed the calls: the selftest adds `add + stream`
    assert a.tolist() == [0, 17, 103, 0] and b.tolist() == [1007, 0, 0, 0]
    ncalls, npatches, names, streams = fast.tape_info(tape)
    assert ncalls == 3 and npatches == 2 and names == ["osp_tape_selftest"] * 3 and streams == [None, 3, None]
    # replay against a DIFFERENT input buffer, on a different current stream
    c = np.zeros(4, dtype=np.int64)
    assert fast.tape_replay(tape, [_addr(c)], 20) == 0
    assert c.tolist() == [0, 30, 103, 0]                       # 10 + stream 20; 100 + its own stream 3
    assert a.tolist() == [0, 17, 103, 0]                       # the original input is untouched by the replay
    assert b.tolist() == [2027, 0, 0, 0]                       # the non-input buffer was written again (1000 + 20)
    assert fast.tape_replay(tape, [_addr(a)], 7) == 0
    assert a.tolist() == [0, 34, 206, 0]
    with pytest.raises(ValueError):
        fast.tape_replay(tape, [], 7)                          # wrong number of inputs


def test_host_tables_are_copied_and_their_addresses_patched(fast):
    L = _lib.lib()
    idx = L._fidx["osp_tape_selftest_table"]
    src = np.array([5, 6, 7, 8], dtype=np.int64)               # "input": the table points at its elements
    out = np.zeros(1, dtype=np.int64)
    table = np.array([_addr(src, 0), _addr(src, 3)], dtype=np.int64)
    fast.tape_begin([_addr(src), _addr(out)], [src.nbytes, out.nbytes], 0)
    assert fast.call(idx, 0, table, 2, _addr(out)) == 0
    tape = fast.tape_end()
    assert out[0] == 5 + 8
    table[:] = 0                                               # the tape holds its own copy of the table
    del table
    src2 = np.array([50, 60, 70, 80], dtype=np.int64)
    out2 = np.zeros(1, dtype=np.int64)
    assert fast.tape_replay(tape, [_addr(src2), _addr(out2)], 0) == 0
    assert out2[0] == 50 + 80 and out[0] == 13
    # a bare address for a host table cannot be recorded (the tape could not keep a copy): refused loudly
    t2 = np.array([_addr(src, 1)], dtype=np.int64)
    fast.tape_begin([_addr(src)], [src.nbytes], 0)
    with pytest.raises(RuntimeError, match="bare address"):
        fast.call(idx, 0, t2.ctypes.data, 1, _addr(out))
    fast.tape_abort()
    assert fast.call(idx, 0, t2.ctypes.data, 1, _addr(out)) == 0           # ... but is fine outside a recording
    assert out[0] == 13 + 6


def test_a_failing_call_is_not_recorded_and_replay_reports_the_entry(fast):
    L = _lib.lib()
    idx = L._fidx["osp_tape_selftest"]
    a = np.zeros(2, dtype=np.int64)
    fast.tape_begin([_addr(a)], [a.nbytes], 0)
    assert fast.call(idx, 0, None, 1) != 0                     # null counter: the entry point refuses
    assert fast.call(idx, 0, _addr(a), 1) == 0
    tape = fast.tape_end()
    assert fast.tape_info(tape)[0] == 1
    rc = fast.tape_replay(tape, [0], 0)                        # input address 0 -> the patched pointer is null -> the call fails
    assert isinstance(rc, tuple) and rc[0] != 0 and rc[1] == "osp_tape_selftest"
    with pytest.raises(RuntimeError):
        fast.tape_begin([0], [0], 0)
        fast.tape_begin([0], [0], 0)                           # regions do not nest
    fast.tape_abort()


def test_aten_guard_classifies_operators():
    """The dispatch-mode guard on CPU tensors: host-side arithmetic is ignored, allocation / view operators are known; the set of
    operators the autograd engine itself issues for gradient accumulation is visible to the guard (which is what lets a taped
    backward re-route them)."""
    import torch
    from optispeech_amd import tape as T

    class Rec:
        poisoned, rerouted = None, 0

        def poison(self, why):
            self.poisoned = why
    seen = []

    class Spy(T._AtenGuard):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            seen.append(func._schema.name)
            return super().__torch_dispatch__(func, types, args, kwargs)
    r = Rec()
    x = torch.randn(4, 3, requires_grad=True)
    with Spy(r):
        y = (x.view(3, 4).t() * 2.0)
        z = y + y                                               # two consumers of y: the engine accumulates with an add
        z.sum().backward()
    assert r.poisoned is None                                  # CPU tensors: nothing here launches a device kernel
    assert "aten::view" in seen and "aten::mul" in seen and "aten::add" in seen
    assert all(n in T._NO_KERNEL for n in ("aten::view", "aten::t", "aten::detach", "aten::empty"))
