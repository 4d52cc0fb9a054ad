"""MSI_SEARCH_LEVELS_PER_WAIT (command lists: 12 by default (8 before round 6), at most 16; direct back end: off by default, at most 4):
several cost levels of a graph-based ranking rule are enqueued back to back and collected behind ONE completion wait.
Same answers required: the reference snapshots, random corpora and deadlines replay with 1, 2, 4, 8 and 16 levels per
wait (every other device test runs with the default).  The CPU tier (tests/test_search_hostlogic_cpu.py) holds the host side of this against
the oracle; this file holds the device side."""
import pytest

import tests.test_search_gpu as G

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("per_wait", ["1", "2", "4", "8", "16"])
def test_reference_snapshots_with_levels_per_wait(monkeypatch, per_wait):
    monkeypatch.setenv("MSI_SEARCH_LEVELS_PER_WAIT", per_wait)
    for case in G.CASES:
        G.test_reference_snapshot(case)


@pytest.mark.parametrize("per_wait", ["4", "16"])
def test_random_corpora_with_levels_per_wait(monkeypatch, per_wait):
    monkeypatch.setenv("MSI_SEARCH_LEVELS_PER_WAIT", per_wait)
    G.test_matches_oracle_on_random_corpora(2, 100)
    G.test_matches_oracle_on_random_corpora(3, 3)
    G.test_ranking_score_threshold()
