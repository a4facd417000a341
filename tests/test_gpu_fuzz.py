"""Short runs of the two differential fuzzers (tools/fuzz_api.py, tools/fuzz_hub.py; the long runs are in profiles/r04_fuzz_*.txt):
whatever the call sequence, the library's fast paths give what its plain paths give, and the pipelined hub what the synchronous one."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(7000, 7016))
def test_fast_paths_equal_the_plain_ones_under_random_call_sequences(seed):
    """fused kernel (incl. hop 512 / N > 1), stages side by side on two streams vs two kernels one after the other: the same random
    sequence of 50 C-ABI calls on both contexts, every result and the carried state compared after every run"""
    import fuzz_api
    import supersdr_amd as S
    counts = fuzz_api.one_sequence(S, seed, 50, [])
    assert counts["runs"] > 5


@pytest.mark.parametrize("seed", range(7100, 7110))
def test_pipelined_hub_equals_the_synchronous_hub_under_random_traffic(seed):
    """ragged feeds, blocks, reserve / commit, stalls, parameter / N / display-state changes: same queues, item for item"""
    import fuzz_hub
    import supersdr_amd as S
    from supersdr_amd.workers import IQHub
    compared, _ = (fuzz_hub.one_wire_sequence if seed % 4 == 3 else fuzz_hub.one_sequence)(S, IQHub, seed, 150)      # every fourth: wire hubs
    assert compared > 0
