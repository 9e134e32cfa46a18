"""Host logic of retargetvid_amd.pipeline (no GPU): the per-call flag plan of the streaming clusterer against a direct
simulation of the reference's loop order (smartVidCrop.py:2359-2373: map i is filtered, then blended into map i+1)."""
import numpy as np

from retargetvid_amd import pipeline as PL


def _simulate(n_maps, bnext, batch, max_span):
    """Feeds n_maps through plan_call the way StreamPipeline does; checks the dependency order of every operation and
    that every map is filtered exactly once.  -> number of calls, most rounds of a regular call, most rounds of a flush."""
    states, bn, gid0 = [], [], 0
    filtered, blended_into = set(), set()
    calls, worst_rounds, worst_flush, fed = 0, 0, 0, 0

    def run(flush):
        nonlocal gid0, calls, worst_rounds, worst_flush
        flags, done = PL.plan_call(states, bn, flush)
        calls += 1
        # what svc_cluster_center does with these flags (csrc/svc_tail.hip: plan_rounds): rounds by chain depth
        depth = {}
        for i in range(len(states)):
            if flags[i] & PL.MAP_HELD:
                continue
            prev = flags[i - 1] if i else 0
            if i and (prev & PL.BLEND_NEXT):
                if prev & PL.MAP_HELD:
                    assert gid0 + i - 1 in filtered, 'blend from a held map that is not final'
                    depth[i] = 0
                else:
                    depth[i] = depth[i - 1] + 1
                blended_into.add(gid0 + i)
            else:
                depth[i] = 0
        for i in sorted(depth, key=lambda j: depth[j]):
            g = gid0 + i
            assert g not in filtered, 'map %d filtered twice' % g
            if g and all_bnext[g - 1]:
                assert g - 1 in filtered and g in blended_into, 'map %d filtered before its blend' % g
            filtered.add(g)
        r = max(depth.values()) + 1 if depth else 0
        if flush:
            worst_flush = max(worst_flush, r)
        else:
            worst_rounds = max(worst_rounds, r)
        assert sorted(depth) == sorted(done)
        for i in done:
            states[i] = PL.FINAL
        k = len(states)
        keep = k
        for i in range(k):
            if states[i] == PL.RAW or (bn[i] and (i + 1 == k or states[i + 1] == PL.RAW)):
                keep = i
                break
        gid0 += keep
        del states[:keep], bn[:keep]

    all_bnext = list(bnext)
    while fed < n_maps:
        n = min(batch, n_maps - fed)
        states += [PL.RAW] * n
        bn += all_bnext[fed:fed + n]
        fed += n
        run(len(states) > max_span)
    if any(s == PL.RAW for s in states):
        run(True)
    assert filtered == set(range(n_maps))
    assert blended_into == {i + 1 for i in range(n_maps - 1) if all_bnext[i]}
    return calls, worst_rounds, worst_flush


def test_one_round_per_call_for_cut_chains():
    rng = np.random.RandomState(0)
    for trial in range(40):
        n = int(rng.randint(1, 400))
        bnext = np.zeros(n, bool)
        for c in rng.choice(n, size=max(1, n // 40), replace=False):          # cuts: maps c-1, c, c+1 blend forward
            bnext[max(0, c - 1):c + 2] = True
        bnext[-1] = False
        calls, rounds, flush_rounds = _simulate(n, bnext, 32, 96)
        assert rounds <= 1, (n, rounds)                                        # every regular call is ONE tail round
        assert flush_rounds <= 8                                               # the last call runs the open chains out


def test_degenerate_chains_fall_back_to_rounds():
    n = 200
    calls, rounds, flush_rounds = _simulate(n, [True] * (n - 1) + [False], 32, 96)   # every map chained: cannot be carried
    assert rounds == 1 and flush_rounds > 32
    calls, rounds, flush_rounds = _simulate(n, [False] * n, 32, 96)
    assert rounds == 1 and flush_rounds == 0 and calls == 7


def test_plan_flags_of_the_benchmark_batch():
    # a batch that starts a shot: maps 0 -> 1 -> 2
    flags, done = PL.plan_call([PL.RAW] * 6, [True, True, False, False, False, False])
    assert flags.tolist() == [0, PL.MAP_HELD, PL.MAP_HELD, 0, 0, 0] and done == [0, 3, 4, 5]
    # next call: the span starts at final map 0; map 1 takes its blend and runs, map 2 waits
    flags, done = PL.plan_call([PL.FINAL, PL.RAW, PL.RAW, PL.FINAL, PL.RAW], [True, True, False, False, False])
    assert flags.tolist() == [PL.MAP_HELD | PL.BLEND_NEXT, 0, PL.MAP_HELD, PL.MAP_HELD, 0] and done == [1, 4]
