"""The role assignment of the persistent stage kernels (csrc/stage_common.h: stage_ticket), restated on the host and run under ARBITRARY dispatch orders and workgroup -> XCD
placements -- the two things HIP does not promise (MI355X_MICROARCH.md, "Workgroup dispatch": contract) and the kernels of round 4 leaned on.  Checked, for thousands of random
placements: every workgroup gets exactly one (slot, role); at any moment of the dispatch the started workgroups form complete slots plus at most one incomplete slot per counter; and
with `capacity >= 8 (NWG - 1) + 1` resident workgroups a dispatcher that only starts a workgroup when a resident slot has finished never stalls (the progress bound behind
lmv_*stage_max_concurrent).  Pure Python: the GPU tests (test_{s,d}stage_roles_by_ticket_under_foreign_placement) check the device code against unskewed runs."""
import random

import pytest


def take_ticket(counters, quota, xcc, skew_hash=0):
    """stage_ticket: ask the counter of the XCD the workgroup runs on, fall through to the next ones when it is used up."""
    for k in range(8):
        y = (xcc + skew_hash + k) & 7
        t = counters[y]
        counters[y] += 1          # (the atomic fetch-add happens even when the ticket is beyond the quota)
        if t < quota:
            return y, t
    return None


@pytest.mark.parametrize("nwg,nslots", [(2, 128), (2, 256), (8, 64), (29, 32), (15, 64), (7, 32), (97, 8), (25, 16)])
def test_every_workgroup_gets_one_role_under_any_placement(nwg, nslots):
    rng = random.Random(nwg * 1000 + nslots)
    quota = nslots // 8 * nwg
    grid = 8 * quota
    for trial in range(40):
        mode = trial % 4
        counters = [0] * 8
        seen = set()
        partial_max = 0
        filled = {}
        for b in range(grid):
            if mode == 0:
                xcc = b % 8                                   # the round-robin placement observed on MI355X
            elif mode == 1:
                xcc = rng.randrange(8)                        # arbitrary
            elif mode == 2:
                xcc = 0 if b < grid // 2 else rng.randrange(8)   # half of the grid lands on one XCD
            else:
                xcc = (b // 37) % 8                           # runs of 37 on one XCD
            got = take_ticket(counters, quota, xcc)
            assert got is not None, (mode, b)
            y, t = got
            slot, role = (t // nwg) * 8 + y, t % nwg
            assert 0 <= slot < nslots and (slot, role) not in seen
            seen.add((slot, role))
            filled[slot] = filled.get(slot, 0) + 1
            # the started workgroups: complete slots + at most one incomplete slot per counter
            incomplete = [s for s, n in filled.items() if n < nwg]
            assert len(incomplete) <= 8 and len({s % 8 for s in incomplete}) == len(incomplete)
            partial_max = max(partial_max, sum(filled[s] for s in incomplete))
        assert len(seen) == grid
        assert partial_max <= 8 * (nwg - 1)


@pytest.mark.parametrize("nwg,nslots,launches", [(29, 32, 4), (8, 64, 9), (2, 128, 31), (15, 64, 4)])
def test_progress_with_the_documented_capacity_bound(nwg, nslots, launches):
    """`launches` concurrent launches of one shape on a device that holds exactly launches * 8 (NWG - 1) + 1 workgroups: a slot runs to its end only when complete; the dispatcher
    starts queued workgroups (of any launch, in random order) whenever there is room.  Must drain without ever being stuck with a full device of incomplete slots."""
    rng = random.Random(7 * nwg + launches)
    capacity = launches * 8 * (nwg - 1) + 1
    quota = nslots // 8 * nwg
    for trial in range(6):
        counters = [[0] * 8 for _ in range(launches)]
        queues = [list(range(8 * quota)) for _ in range(launches)]
        filled = [dict() for _ in range(launches)]
        resident, done = 0, 0
        total = launches * 8 * quota
        guard = 0
        while done < total:
            guard += 1
            assert guard < 20 * total, "no progress"
            # complete slots finish and free their workgroups
            for l in range(launches):
                for s, n in list(filled[l].items()):
                    if n == nwg:
                        del filled[l][s]
                        resident -= nwg
                        done += nwg
            # the dispatcher starts workgroups while there is room, from randomly chosen launches, on randomly chosen XCDs
            started = False
            while resident < capacity:
                cand = [l for l in range(launches) if queues[l]]
                if not cand:
                    break
                l = rng.choice(cand)
                queues[l].pop()
                y, t = take_ticket(counters[l], quota, rng.randrange(8))
                slot = (t // nwg) * 8 + y
                filled[l][slot] = filled[l].get(slot, 0) + 1
                resident += 1
                started = True
            stuck = done < total and not started and not any(n == nwg for f in filled for n in f.values())
            assert not stuck, f"device full of incomplete slots: {resident} resident of {capacity}"
