"""The filter oracle (oracle/filter_oracle.py, a restatement of search/facet/filter/index_filter.rs) pinned to the
46 cases of the reference's own filter tests (crates/milli/tests/search/filters.rs over test_set.ndjson) that its
helper covers at top level: tests/golden/filter_fixtures.json."""
import json
import os

from oracle import filter_oracle as FO
from tests.toy_filter import parse
from tests.toy_milli import ToyMilli

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = json.load(open(os.path.join(ROOT, "tests", "golden", "filter_fixtures.json")))


def tree_of(groups):
    return ("and", [("or", [parse(f) for f in grp]) for grp in groups])


def test_oracle_replays_the_reference_filter_tests():
    index = ToyMilli(FIX["docs"], searchable=["title", "description"])
    assert len(FIX["cases"]) >= 45
    for case in FIX["cases"]:
        got = FO.evaluate(index, tree_of(case["groups"]))
        assert sorted(index.docs[d]["id"] for d in got) == case["ids"], case["name"]


def test_parser_covers_the_grammar_the_tests_use():
    assert parse("tag=red") == ("cond", "tag", "=", ["red"])
    assert parse("NOT opt1 IS NOT NULL") == ("not", ("not", ("cond", "opt1", "null", [])))
    assert parse("price 10 TO 20.5") == ("cond", "price", "to", ["10", "20.5"])
    assert parse("tag_in NOT IN[1, 2, four]") == ("not", ("cond", "tag_in", "in", ["1", "2", "four"]))
    assert parse("title STARTS WITH 'hell o'") == ("cond", "title", "startswith", ["hell o"])
    assert parse("_geoBoundingBox([45.5, 9.3], [45.4, 9.1])") == ("geo_bbox", (45.5, 9.3), (45.4, 9.1))
    assert parse("(a = 1 OR b > 2) AND NOT c EXISTS") == (
        "and", [("or", [("cond", "a", "=", ["1"]), ("cond", "b", ">", ["2"])]), ("not", ("cond", "c", "exists", []))])


def test_distance_formula_against_the_reference_dataset():
    """test_set.ndjson carries `geo_rank` = the distance in metres of every document from the point the geo tests use
    (filters.rs: `_geoRadius(50.630010347667806, 3.086251829166809, ...)` vs `geo_rank < 100000`), rounded up: the
    restated geoutils haversine (third-party, from memory of the crate) reproduces all 17 — 43 m to 9 499 586 m."""
    import math
    from oracle.ranking_oracle import distance_between_two_points
    base = (50.630010347667806, 3.086251829166809)
    n = 0
    for d in FIX["docs"]:
        if "_geo" in d:
            assert math.ceil(distance_between_two_points(base, (d["_geo"]["lat"], d["_geo"]["lng"]))) == d["geo_rank"], d["id"]
            n += 1
    assert n == 17


def test_distance_formula_against_the_geo_distance_literals_of_the_http_tests():
    """`_geoDistance` = distance_between_two_points(...).round() (crates/meilisearch/src/search/mod.rs:2786); literals of
    crates/meilisearch/tests/search/geo.rs:101,113,287-305 and tests/documents/add_documents.rs:1865-2067."""
    from oracle.ranking_oracle import distance_between_two_points as dist
    for a, b, want in [((45.4777599, 9.1967508), (45.4777599, 9.1967508), 0),
                       ((45.4777599, 9.1967508), (34.0522, -118.2437), 9714063),
                       ((0.0, 0.0), (-89.0, 0.0), 9896348), ((0.0, 0.0), (0.0, 178.0), 19792697),
                       ((50.629973371633746, 3.0569447399419567), (1.0, 1.0), 5522018),
                       ((50.629973371633746, 3.0569447399419567), (2.0, 2.0), 5408322),
                       ((10.0, 0.0), (4.0, 0.0), 667170), ((10.0, 0.0), (3.0, 0.0), 778364), ((10.0, 0.0), (5.0, 0.0), 555975)]:
        assert int(dist(a, b) + 0.5) == want, (a, b)
