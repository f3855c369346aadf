"""Placement of the training step's HIP streams on the GPU's hardware queues ("lanes").

Measured in round 4: the ROCm 7.2 runtime multiplexes every HIP stream of a process onto GPU_MAX_HW_QUEUES = 4 hardware
queues, dealt out round-robin; work of two streams that share a queue executes in enqueue order, one after the other, whenever the
host enqueues faster than the device drains (which the call tapes make the normal case).  The step uses ~16 streams -- the calling
stream, the vocoder stream, the CTC side stream, two weight-gradient side streams, eight sub-discriminator streams, the spectral-loss
stream, the discriminator-phase calling stream -- so WHICH of them share a queue decides how much of the multi-stream schedule is
real concurrency: shifting the assignment by one stream moved the step between 17.2 and 19.7 ms (profiles/r04_stream_skew.txt).

This module makes the assignment explicit and deterministic: a pool of streams is created up front, in a fixed order, and each is
bound to its hardware queue right away (one trivial launch); pool entry i sits on lane i mod 4.  A logical stream asks for a lane by
name; ``OSP_LANES="name:lane,..."`` overrides the table (tools/lane_search.py (git history) uses that).  Names: voc, ctc, wg_main, wg_voc, wg_other,
p0..p4 (period discriminators 2, 3, 5, 7, 11), r0..r2 (resolution discriminators), spec, dphase.
"""
import os

import torch

N_LANES = int(os.environ.get("OSP_N_LANES", "4"))      # = GPU_MAX_HW_QUEUES of the process (4 unless the environment says otherwise)
DEPTH = 7                      # pool entries per lane

#: lane per logical stream (None = an unmanaged stream from torch's pool, as before round 4).  The table is the result of two searches.
#: First: 36 random balanced deals plus 30 one / two-stream mutations of the winner, each candidate = one `bench.py` run: 16.65-16.81 ms /
#: step against 17.5-17.7 for the unmanaged assignment on the same boxes; the 70 candidates span 16.7 ... 20.3 ms
#: (profiles/r04_lane_search.txt).  Second (tools/lane_search2.py, 80-120 pipelined steps per candidate, all on one box): 60 mutations
#: + 30 random deals, then 45 mutations around each new winner until none improved: 15.9 ms / step against 16.3 for the first
#: table, re-measured three times each on two boxes (profiles/r04_lane_search{2,3,4}.txt).  OSP_LANES="voc:1,..." overrides entries,
#: OSP_LANES_OFF=1 gives the unmanaged streams back.
DEFAULT = {"voc": 0, "ctc": 1, "wg_main": 1, "wg_voc": 0, "wg_other": None, "p0": 0, "p1": 3, "p2": 0, "p3": 3,
           "p4": 2, "r0": 0, "r1": 1, "r2": 2, "spec": 1, "dphase": 1}

_pool = {}
_taken = {}
_table = None


def table():
    global _table
    if _table is None:
        t = dict(DEFAULT)
        if os.environ.get("OSP_LANES_OFF", "0") == "1":
            t = {k: None for k in t}
        for item in os.environ.get("OSP_LANES", "").split(","):
            if ":" in item:
                k, v = item.split(":")
                t[k.strip()] = None if v.strip() in ("-", "") else int(v)
        _table = t
    return _table


def managed():
    return any(v is not None for v in table().values())


def _make_pool(device):
    from . import kernels as K
    idx = torch.device(device).index
    idx = torch.cuda.current_device() if idx is None else idx
    if idx not in _pool:
        streams = []
        scratch = torch.zeros(1, dtype=torch.int64, device=device)
        for i in range(N_LANES * DEPTH):
            s = torch.cuda.Stream(device=device)
            with torch.cuda.stream(s):
                K.store_i64(scratch, i)                      # first use binds the stream to its hardware queue: do it in pool order
            streams.append(s)
        # torch deals streams from a pool of 32 per priority: past that, requests ALIAS pool entries and two logical lanes would
        # silently share one HIP stream.  28 are taken here; say so if the process had already used up the pool.
        if len({st.cuda_stream for st in streams}) != len(streams):
            import warnings
            warnings.warn("optispeech_amd.lanes: torch's stream pool is exhausted, lane streams alias each other -- the measured "
                          "stream -> hardware-queue table does not hold in this process")
        torch.cuda.synchronize(device)
        _pool[idx], _taken[idx] = streams, set()
    return idx


def stream(name, device):
    """The persistent stream of logical name ``name`` (a fresh one per call: callers cache it)."""
    lane = table().get(name)
    if lane is None:
        return torch.cuda.Stream(device=device)
    idx = _make_pool(device)
    for i, s in enumerate(_pool[idx]):
        if i % N_LANES == lane % N_LANES and i not in _taken[idx]:
            _taken[idx].add(i)
            return s
    return torch.cuda.Stream(device=device)
