"""TEST INFRASTRUCTURE — CPU restatement of the evaluation of a filter over the facet databases
(crates/milli/src/search/facet/filter/index_filter.rs:84-340,344-460,465-690; value_bounds.rs:20-90), over the toy
index of tests/toy_milli.py, with plain Python sets.  Never imported by the product.

The expression arrives as the tree tests/toy_filter.py parses:
  ("and", [e...]) ("or", [e...]) ("not", e)
  ("cond", field, op, args)   op in = != > >= < <= to exists null empty in startswith contains
  ("geo_radius", lat, lng, radius) ("geo_bbox", (top, right), (bottom, left))
Pinned by the reference's own filter tests (crates/milli/tests/search/filters.rs over tests/assets/test_set.ndjson,
expected ids from a restatement of its execute_filter helper: tests/golden/filter_fixtures.json)."""
import math

from oracle.ranking_oracle import distance_between_two_points


def parse_finite_float(tok):
    try:
        x = float(tok)
    except ValueError:
        return None
    return x if math.isfinite(x) else None


def normalize_facet(s):
    import unicodedata
    return unicodedata.normalize("NFKD", s.strip()).lower()


def _numbers_in(index, field, lo, lo_incl, hi, hi_incl):
    """explore_facet_levels over facet_id_f64_docids: empty when the bounds cross (index_filter.rs:315-321)."""
    if lo > hi or (lo == hi and not (lo_incl and hi_incl)):
        return set()
    out = set()
    for d, vals in enumerate(index.facet_numbers(field)):
        if any((x > lo or (lo_incl and x == lo)) and (x < hi or (hi_incl and x == hi)) for x in vals):
            out.add(d)
    return out


def _strings_where(index, field, pred):
    per_doc, values = index.facet_strings(field)
    ok = {i for i, v in enumerate(values) if pred(v.encode())}
    return {d for d, ranks in enumerate(per_doc) if any(r in ok for r in ranks)}


def evaluate_condition(index, field, op, args):
    """evaluate_operator, index_filter.rs:84-253 (the feature checks are the shim's)."""
    fmax = 1.7976931348623157e308
    if op in (">", ">=", "<", "<=", "to"):
        out = set()
        if op == "to":
            a, b = parse_finite_float(args[0]), parse_finite_float(args[1])
            if a is not None and b is not None:
                out |= _numbers_in(index, field, a, True, b, True)
            lo, hi = normalize_facet(args[0]).encode(), normalize_facet(args[1]).encode()
            if lo <= hi:
                out |= _strings_where(index, field, lambda v: lo <= v <= hi)
            return out
        x = parse_finite_float(args[0])
        s = normalize_facet(args[0]).encode()
        if op == ">":
            out |= _numbers_in(index, field, x, False, fmax, True) if x is not None else set()
            out |= _strings_where(index, field, lambda v: v > s)
        elif op == ">=":
            out |= _numbers_in(index, field, x, True, fmax, True) if x is not None else set()
            out |= _strings_where(index, field, lambda v: v >= s)
        elif op == "<":
            out |= _numbers_in(index, field, -fmax, True, x, False) if x is not None else set()
            out |= _strings_where(index, field, lambda v: v < s)
        else:
            out |= _numbers_in(index, field, -fmax, True, x, True) if x is not None else set()
            out |= _strings_where(index, field, lambda v: v <= s)
        return out
    if op == "exists":
        return index.exists_docids(field)
    if op == "null":
        return index.null_docids(field)
    if op == "empty":
        return index.empty_docids(field)
    if op in ("=", "!="):
        x = parse_finite_float(args[0])
        s = normalize_facet(args[0]).encode()
        eq = _strings_where(index, field, lambda v: v == s)       # evaluate_equal, value_bounds.rs:94-119
        if x is not None:
            eq |= _numbers_in(index, field, x, True, x, True)
        return eq if op == "=" else index.all_docids() - eq
    if op == "in":                                                # index_filter.rs:367-391: OR of Equal
        out = set()
        for el in args:
            out |= evaluate_condition(index, field, "=", [el])
        return out
    if op == "contains":
        s = normalize_facet(args[0]).encode()
        return _strings_where(index, field, lambda v: s in v)
    if op == "startswith":                                        # index_filter.rs:198-250
        s = normalize_facet(args[0]).encode()
        if not s:
            return index.exists_docids(field)
        return _strings_where(index, field, lambda v: v.startswith(s))
    raise ValueError(op)


def evaluate(index, e):
    """inner_evaluate, index_filter.rs:344-460 (no universe hint: the result is the same set, the hint only prunes)."""
    k = e[0]
    if k == "not":
        return index.all_docids() - evaluate(index, e[1])
    if k == "or":
        out = set()
        for x in e[1]:
            out |= evaluate(index, x)
        return out
    if k == "and":
        out = None
        for x in e[1]:
            r = evaluate(index, x)
            out = r if out is None else out & r
        return out if out is not None else index.all_docids()
    if k == "cond":
        return evaluate_condition(index, e[1], e[2], e[3])
    if k == "geo_radius":                                         # :465-503 (the R-tree walk stops at the first farther point)
        _, lat, lng, radius = e
        eps = 2.220446049250313e-16
        return {d for d, pt in index.geo_points.items() if distance_between_two_points((lat, lng), pt) <= radius + eps}
    if k == "geo_bbox":                                           # :531-660: Between on _geo.lat and _geo.lng
        _, (top, right), (bottom, left) = e
        lat = evaluate_condition(index, "_geo.lat", "to", [repr(bottom), repr(top)])
        if right < left:
            lng = (evaluate_condition(index, "_geo.lng", "to", [repr(left), "180.0"])
                   | evaluate_condition(index, "_geo.lng", "to", ["-180.0", repr(right)]))
        else:
            lng = evaluate_condition(index, "_geo.lng", "to", [repr(left), repr(right)])
        return lat & lng
    raise ValueError(k)


def vector_filter(kind, has_fragments, store_items, user_provided, skip_regenerate):
    """evaluate_inner of the `_vectors` filter for one embedder, on plain sets
    (crates/milli/src/search/facet/filter/vector.rs:78-158).  store_items: the docid sets of the stores the variant
    looks at (Fragment: the fragment's store, :101-125; the others: every store of the embedder, i.e. what
    VectorStore::aggregate_stats().documents holds)."""
    documents = set().union(*store_items) if store_items else set()
    if kind == "fragment":
        return documents - set(user_provided)                      # :121-124
    if kind == "documentTemplate":
        return set() if has_fragments else documents - set(user_provided)   # :126-135
    if kind == "userProvided":
        return set(user_provided)                                  # :136-139
    if kind == "regenerate":
        return documents - set(skip_regenerate)                    # :140-145
    return documents                                               # :146-150


def vector_filter_all(embedders, kind, universe=None):
    """evaluate (vector.rs:49-76): the union over the named embedders, then `& universe`.
    embedders: [(has_fragments, store_items, user_provided, skip_regenerate)]."""
    out = set()
    for has_fragments, store_items, up, sr in embedders:
        out |= vector_filter(kind, has_fragments, store_items, up, sr)
    return out & set(universe) if universe is not None else out
