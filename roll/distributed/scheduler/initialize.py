"""``init()`` of the reference starts / attaches a Ray cluster (roll/distributed/scheduler/initialize.py:54-83).
Here the data-parallel workers are the torchrun processes themselves (one per GPU), so init() only joins the
process group when launched with WORLD_SIZE > 1 and is a no-op otherwise."""
from socioreasoner_amd import dp


def init():
    return dp.init_distributed()
