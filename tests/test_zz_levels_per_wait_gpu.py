"""MSI_SEARCH_LEVELS_PER_WAIT (default 1 = off): several cost levels of a graph-based ranking rule are enqueued back
to back (msi_bits_paths_enqueue, one counts region per level) and collected behind ONE completion wait
(msi_bits_paths_collect).  Same answers required: the reference snapshots, random corpora and deadlines replay with
2 and 4 levels per wait.  The CPU tier (tests/test_search_hostlogic_cpu.py) holds the host side of this against
the oracle; this file holds the device side.  Runs last on purpose (experimental knob, off by default)."""
import pytest

import tests.test_search_gpu as G

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("per_wait", ["2", "4"])
def test_reference_snapshots_with_levels_per_wait(monkeypatch, per_wait):
    monkeypatch.setenv("MSI_SEARCH_LEVELS_PER_WAIT", per_wait)
    for case in G.CASES:
        G.test_reference_snapshot(case)


def test_random_corpora_with_levels_per_wait(monkeypatch):
    monkeypatch.setenv("MSI_SEARCH_LEVELS_PER_WAIT", "4")
    G.test_matches_oracle_on_random_corpora(2, 100)
    G.test_matches_oracle_on_random_corpora(3, 3)
    G.test_ranking_score_threshold()
