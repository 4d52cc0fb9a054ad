"""TEST INFRASTRUCTURE — the part of the shim that walks a filter: a parser for the filter strings the reference's
tests use (a subset of filter-parser's grammar) and DeviceFilter, which evaluates the tree with the product's leaf
kernels (msi_bits_facet_range / _facet_in / _geo_within) and msi_bits_op, the way the Rust shim would from
IndexFilter::inner_evaluate (crates/milli/src/search/facet/filter/index_filter.rs:344-460)."""
import re

import numpy as np

from oracle.filter_oracle import normalize_facet, parse_finite_float


def _split_top(s, sep):
    out, depth, cur, i = [], 0, "", 0
    while i < len(s):
        c = s[i]
        depth += c in "([" 
        depth -= c in ")]"
        if depth == 0 and s.startswith(sep, i):
            out.append(cur)
            cur = ""
            i += len(sep)
            continue
        cur += c
        i += 1
    out.append(cur)
    return out


def parse(f):
    """-> the tree oracle/filter_oracle.py documents."""
    f = f.strip()
    parts = _split_top(f, " OR ")
    if len(parts) > 1:
        return ("or", [parse(p) for p in parts])
    parts = _split_top(f, " AND ")
    if len(parts) > 1:
        return ("and", [parse(p) for p in parts])
    if f.startswith("NOT "):
        return ("not", parse(f[4:]))
    if f.startswith("(") and f.endswith(")"):
        return parse(f[1:-1])
    m = re.fullmatch(r"_geoRadius\(\s*([-\d.e]+)\s*,\s*([-\d.e]+)\s*,\s*([-\d.e]+)\s*\)", f)
    if m:
        return ("geo_radius", float(m.group(1)), float(m.group(2)), float(m.group(3)))
    m = re.fullmatch(r"_geoBoundingBox\(\s*\[\s*([-\d.e]+)\s*,\s*([-\d.e]+)\s*\]\s*,\s*\[\s*([-\d.e]+)\s*,\s*([-\d.e]+)\s*\]\s*\)", f)
    if m:
        return ("geo_bbox", (float(m.group(1)), float(m.group(2))), (float(m.group(3)), float(m.group(4))))

    def val(v):
        v = v.strip()
        return v[1:-1] if len(v) >= 2 and v[0] == v[-1] and v[0] in "'\"" else v
    for pat, build in (
        (r"(\S+)\s+NOT\s+IN\s*\[(.*)\]", lambda m: ("not", ("cond", m.group(1), "in", [val(x) for x in m.group(2).split(",")]))),
        (r"(\S+)\s+IN\s*\[(.*)\]", lambda m: ("cond", m.group(1), "in", [val(x) for x in m.group(2).split(",")])),
        (r"(\S+)\s+NOT\s+EXISTS", lambda m: ("not", ("cond", m.group(1), "exists", []))),
        (r"(\S+)\s+EXISTS", lambda m: ("cond", m.group(1), "exists", [])),
        (r"(\S+)\s+IS\s+NOT\s+NULL", lambda m: ("not", ("cond", m.group(1), "null", []))),
        (r"(\S+)\s+IS\s+NULL", lambda m: ("cond", m.group(1), "null", [])),
        (r"(\S+)\s+IS\s+NOT\s+EMPTY", lambda m: ("not", ("cond", m.group(1), "empty", []))),
        (r"(\S+)\s+IS\s+EMPTY", lambda m: ("cond", m.group(1), "empty", [])),
        (r"(\S+)\s+NOT\s+STARTS\s+WITH\s+(.*)", lambda m: ("not", ("cond", m.group(1), "startswith", [val(m.group(2))]))),
        (r"(\S+)\s+STARTS\s+WITH\s+(.*)", lambda m: ("cond", m.group(1), "startswith", [val(m.group(2))])),
        (r"(\S+)\s+NOT\s+CONTAINS\s+(.*)", lambda m: ("not", ("cond", m.group(1), "contains", [val(m.group(2))]))),
        (r"(\S+)\s+CONTAINS\s+(.*)", lambda m: ("cond", m.group(1), "contains", [val(m.group(2))])),
        (r"(\S+?)\s+(\S+)\s+TO\s+(\S+)", lambda m: ("cond", m.group(1), "to", [val(m.group(2)), val(m.group(3))])),
        (r"([^\s!<>=]+)\s*(!=|>=|<=|=|>|<)\s*(.*)", lambda m: ("cond", m.group(1), m.group(2), [val(m.group(3))])),
    ):
        m = re.fullmatch(pat, f)
        if m:
            return build(m)
    raise ValueError(f"cannot parse filter: {f}")


FMAX = 1.7976931348623157e308


class DeviceFilter:
    """Evaluates a filter tree on the device.  Per (field, kind) one FacetKeys table, staged on first use (the shim
    stages them per `updated_at`); string predicates are resolved against the field's sorted distinct values on the
    host (the shim: facet_id_string_fst), the documents are selected on the device."""

    def __init__(self, ma, ctx, index, n_slots=48):
        self.ma, self.ctx, self.index = ma, ctx, index
        self.pool = ma.BitsPool(ctx, max(index.n_docs, 1), n_slots)
        self.free = list(range(n_slots - 1, 0, -1))
        self.pool.fill(0, True)                      # slot 0: documents_ids
        self.numbers, self.strings, self.points = {}, {}, None

    def close(self):
        for t in list(self.numbers.values()) + [s[0] for s in self.strings.values()]:
            t.close()
        if self.points is not None:
            self.points.close()
        self.pool.close()

    def _numbers(self, field):
        if field not in self.numbers:
            per_doc = [[self.ma.facet_number_key(x) for x in vals] for vals in self.index.facet_numbers(field)]
            self.numbers[field] = self.ma.FacetKeys(self.ctx, per_doc)
        return self.numbers[field]

    def _strings(self, field):
        if field not in self.strings:
            per_doc, values = self.index.facet_strings(field)
            self.strings[field] = (self.ma.FacetKeys(self.ctx, per_doc), [v.encode() for v in values])
        return self.strings[field]

    def _number_range(self, field, dst, lo, lo_incl, hi, hi_incl, accumulate):
        k = self.ma.facet_number_key
        a, b = k(lo) + (0 if lo_incl else 1), k(hi) - (0 if hi_incl else 1)
        if lo > hi:
            a, b = 1, 0
        self.pool.facet_range(self._numbers(field), a, b if b >= 0 else 0, dst, accumulate)

    def _string_ranks(self, field, dst, ranks, accumulate):
        keys, _ = self._strings(field)
        ranks = sorted(ranks)
        if ranks and ranks == list(range(ranks[0], ranks[-1] + 1)):
            self.pool.facet_range(keys, ranks[0], ranks[-1], dst, accumulate)      # an interval of the value order
        else:
            self.pool.facet_in(keys, ranks, dst, accumulate)

    def _cond(self, field, op, args, dst):
        p = self.pool
        values = self._strings(field)[1] if op not in ("exists", "null", "empty") else None
        if op in ("exists", "null", "empty"):
            docs = {"exists": self.index.exists_docids, "null": self.index.null_docids, "empty": self.index.empty_docids}[op](field)
            p.set_from_docids(dst, np.array(sorted(docs), dtype=np.uint32))      # the index's own bitmaps
            return
        if op in (">", ">=", "<", "<=", "to"):
            p.fill(dst, False)
            if op == "to":
                a, b = parse_finite_float(args[0]), parse_finite_float(args[1])
                if a is not None and b is not None:
                    self._number_range(field, dst, a, True, b, True, True)
                lo, hi = normalize_facet(args[0]).encode(), normalize_facet(args[1]).encode()
                self._string_ranks(field, dst, [i for i, v in enumerate(values) if lo <= v <= hi], True)
                return
            x, s = parse_finite_float(args[0]), normalize_facet(args[0]).encode()
            if x is not None:
                if op in (">", ">="):
                    self._number_range(field, dst, x, op == ">=", FMAX, True, True)
                else:
                    self._number_range(field, dst, -FMAX, True, x, op == "<=", True)
            cmp_ = {">": lambda v: v > s, ">=": lambda v: v >= s, "<": lambda v: v < s, "<=": lambda v: v <= s}[op]
            self._string_ranks(field, dst, [i for i, v in enumerate(values) if cmp_(v)], True)
            return
        if op in ("=", "!=", "in"):
            p.fill(dst, False)
            for el in (args if op == "in" else args[:1]):
                x, s = parse_finite_float(el), normalize_facet(el).encode()
                self._string_ranks(field, dst, [i for i, v in enumerate(values) if v == s], True)
                if x is not None:
                    self._number_range(field, dst, x, True, x, True, True)
            if op == "!=":
                p.op(dst, 0, dst, 2)                 # documents_ids - equal
            return
        if op == "contains":
            s = normalize_facet(args[0]).encode()
            self._string_ranks(field, dst, [i for i, v in enumerate(values) if s in v], False)
            return
        if op == "startswith":
            s = normalize_facet(args[0]).encode()
            if not s:
                p.set_from_docids(dst, np.array(sorted(self.index.exists_docids(field)), dtype=np.uint32))
                return
            self._string_ranks(field, dst, [i for i, v in enumerate(values) if v.startswith(s)], False)
            return
        raise ValueError(op)

    def _eval(self, e):
        """-> slot holding the documents of e (the caller frees it)"""
        p, k = self.pool, e[0]
        dst = self.free.pop()
        if k == "not":
            s = self._eval(e[1])
            p.op(dst, 0, s, 2)
            self.free.append(s)
        elif k in ("or", "and"):
            p.fill(dst, k == "and")
            for x in e[1]:
                s = self._eval(x)
                p.op(dst, dst, s, 1 if k == "or" else 0)
                self.free.append(s)
        elif k == "cond":
            self._cond(e[1], e[2], e[3], dst)
        elif k == "geo_radius":
            if self.points is None:
                lat_lng = np.full((max(self.index.n_docs, 1), 2), np.nan)
                for d, pt in self.index.geo_points.items():
                    lat_lng[d] = pt
                self.points = self.ma.GeoPoints(self.ctx, lat_lng)
            p.geo_within(self.points, 0, e[1], e[2], e[3], dst)
        elif k == "geo_bbox":
            (top, right), (bottom, left) = e[1], e[2]
            lng = self.free.pop()
            p.fill(dst, False)
            self._number_range("_geo.lat", dst, bottom, True, top, True, True)
            p.fill(lng, False)
            if right < left:
                self._number_range("_geo.lng", lng, left, True, 180.0, True, True)
                self._number_range("_geo.lng", lng, -180.0, True, right, True, True)
            else:
                self._number_range("_geo.lng", lng, left, True, right, True, True)
            p.op(dst, dst, lng, 0)
            self.free.append(lng)
        else:
            raise ValueError(k)
        return dst

    def evaluate(self, e):
        s = self._eval(e)
        out = self.pool.to_docids(s).tolist()
        self.free.append(s)
        return out
