"""Counter-based RNG bookkeeping for the fused dropout sites (Philox key = (seed, site stream id)).

Every dropout site owns a fixed `stream id`; the seed advances once per training step, so forward and
backward of a step regenerate the same masks from (seed, stream, element index) without storing them.
Data-parallel ranks offset the seed (rank * 2**32) so their masks are independent.

Stream ids are handed out per model: ``reset_streams()`` is called by ``config.make_optispeech`` before the modules are
constructed, so the ids depend only on the construction order inside ONE model -- a model rebuilt in the same process
(checkpoint resume) draws the same masks as the saved run.

hipGraph replay: a captured step cannot take the seed as a kernel argument (it would be frozen at capture time), so while
a step is captured / replayed ``seed()`` returns a ``DeviceSeed`` -- the kernels then read the seed from device memory
(``seed_dev`` of the C ABI) and the replay loop writes the new seed there before each launch (``set_device_seed``).
"""
_state = {"seed": 1234, "next_stream": 1}
_dev = {"seed": None}


class DeviceSeed:
    """The per-step seed as a one-element int64 device tensor (read by the kernels through ``seed_dev``)."""
    __slots__ = ("tensor",)

    def __init__(self, tensor):
        self.tensor = tensor


def manual_seed(seed: int, rank: int = 0):
    _state["seed"] = int(seed) + (int(rank) << 32)


def advance():
    _state["seed"] += 1


def host_seed() -> int:
    return _state["seed"]


def seed():
    """What a dropout site passes to its kernel: the integer seed, or the DeviceSeed while a graph is captured."""
    return _dev["seed"] if _dev["seed"] is not None else _state["seed"]


def use_device_seed(tensor):
    """tensor: int64[1] on the device (None switches back to host seeds)."""
    _dev["seed"] = DeviceSeed(tensor) if tensor is not None else None


def device_seed_active() -> bool:
    """True while dropout sites hand their kernels the seed as a device tensor (graph capture / replay, taped steps)."""
    return _dev["seed"] is not None


def new_stream() -> int:
    s = _state["next_stream"]
    _state["next_stream"] += 1
    return s


def reset_streams():
    _state["next_stream"] = 1
