#!/usr/bin/env python3
"""Step-level model of the continuous batcher on bench.py's ragged workload (128 requests, max_new uniform in [64, 192], seed 4000, 32 rows): how
many decode steps the run takes as a function of the scheduling round, the number of staged groups the engine could hold, the spare KV slots and
the admission speed.  Reproduces the measured 676 steps (672) at 4 steps per round and says where the rest of the gap to an ideal refill is.
CPU only:  python tools/sim_ragged.py"""
from collections import deque

import numpy as np

LENS = np.random.default_rng(4000).integers(64, 193, size=128).tolist()


def run(poll=4, banks=1, rows=32, spare=32, adm_steps_per_request=2.85):
    """adm_steps_per_request: decode steps that pass while ONE request is prefilled on the admission share of the chip (measured: 289 ms per
    32 requests beside 3.17 ms steps).  A staged group is usable when all of it is prefilled; `banks` groups may be staged at a time."""
    pending, staged_lens, t, steps, busy_until = deque(LENS), deque(), 0, 0, 0
    active = [pending.popleft() for _ in range(rows)]
    free_rows, slots, staged = [], spare, deque()
    while any(a is not None for a in active) or pending or staged:
        while free_rows and staged and staged[0][0] <= t:
            staged[0][1] -= 1
            active[free_rows.pop()] = staged_lens.popleft()
            if staged[0][1] == 0:
                staged.popleft()
        if len(staged) < banks and pending and slots > 0:
            k = min(slots, len(pending))
            busy_until = max(t, busy_until) + int(np.ceil(adm_steps_per_request * k))
            for _ in range(k):
                staged_lens.append(pending.popleft())
            staged.append([busy_until, k])
            slots -= k
        t += poll
        steps += poll
        for i, a in enumerate(active):
            if a is not None:
                if a - poll <= 0:
                    active[i] = None
                    free_rows.append(i)
                    slots += 1
                else:
                    active[i] = a - poll
    return steps


if __name__ == "__main__":
    print("ideal (work / rows):", sum(LENS) / 32)
    for poll in (16, 8, 4, 1):
        print(f"steps per round {poll:2d}: instant admission {run(poll, adm_steps_per_request=0):4d}   measured admission speed {run(poll):4d}")
    print("two / four staged groups at a time:", run(banks=2), run(banks=4))
    print("64 / 96 spare KV slots:", run(spare=64), run(spare=96))
