"""Counter-based RNG bookkeeping for the fused dropout sites (Philox key = (seed, site stream id)).

Every dropout site owns a fixed `stream id`; the seed advances once per training step, so forward and
backward of a step regenerate the same masks from (seed, stream, element index) without storing them.
Data-parallel ranks offset the seed (rank * 2**32) so their masks are independent.
"""
_state = {"seed": 1234, "next_stream": 1}


def manual_seed(seed: int, rank: int = 0):
    _state["seed"] = int(seed) + (int(rank) << 32)


def advance():
    _state["seed"] += 1


def seed() -> int:
    return _state["seed"]


def new_stream() -> int:
    s = _state["next_stream"]
    _state["next_stream"] += 1
    return s
