#!/usr/bin/env python
"""Extracts golden BYTES of a real milli index into tests/golden/index_blobs.json.

The reference ships one LMDB environment written by milli itself (the v1.12 upgrade test,
crates/meilisearch/tests/upgrade/v1_12/v1_12_0.ms/indexes/<uuid>/data.mdb).  It holds what no source file
does: `fst` 0.4.7 blobs (main["words-fst"], main["stop-words"], main["exact-words"], main["words-prefixes-fst"],
facet-id-string-fst[fid]) and CboRoaringBitmap / RoaringBitmap values written by `roaring` 0.10, next to the keys
they were built from.  Only test DATA is extracted (a few hundred bytes); the minimal read-only LMDB page walk
below exists for that and nothing else.  Runs where /root/reference exists (this container); the JSON travels.

    python tests/golden/make_index_fixtures.py
"""
import json
import os
import struct
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

ENV = "/root/reference/crates/meilisearch/tests/upgrade/v1_12/v1_12_0.ms/indexes"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "index_blobs.json")
PAGE = 4096
P_BRANCH, F_BIGDATA, F_SUBDATA = 0x01, 0x01, 0x02


class LmdbFile:
    """Read-only walk of an LMDB data file (lmdb.h / mdb.c layouts: 16-byte page header, u16 node pointers,
    8-byte node header; named databases are F_SUBDATA records of the main database)."""

    def __init__(self, path):
        self.d = open(path, "rb").read()
        metas = []
        for pg in (0, 1):
            off = pg * PAGE + 16
            magic = struct.unpack_from("<I", self.d, off)[0]
            assert magic == 0xBEEFC0DE
            main_db = struct.unpack_from("<IHHQQQQQ", self.d, off + 24 + 48)
            txnid = struct.unpack_from("<Q", self.d, off + 24 + 96 + 8)[0]
            metas.append((txnid, main_db))
        self.main_root = max(metas)[1][7]

    def walk(self, root):
        if root == 0xFFFFFFFFFFFFFFFF:
            return
        off = root * PAGE
        _, _, flags, lower, _ = struct.unpack_from("<QHHHH", self.d, off)
        for i in range((lower - 16) // 2):
            ptr = struct.unpack_from("<H", self.d, off + 16 + 2 * i)[0]
            lo, hi, nflags, ksize = struct.unpack_from("<HHHH", self.d, off + ptr)
            key = self.d[off + ptr + 8: off + ptr + 8 + ksize]
            if flags & P_BRANCH:
                yield from self.walk(lo | (hi << 16) | (nflags << 32))
                continue
            size, at = lo | (hi << 16), off + ptr + 8 + ksize
            if nflags & F_BIGDATA:
                at = struct.unpack_from("<Q", self.d, at)[0] * PAGE + 16
            yield key, self.d[at: at + size], nflags

    def items(self, name):
        for k, v, fl in self.walk(self.main_root):
            if fl & F_SUBDATA and k == name.encode():
                for k2, v2, _ in self.walk(struct.unpack("<IHHQQQQQ", v)[7]):
                    yield k2, v2


def main():
    uuid = sorted(os.listdir(ENV))[0]
    env = LmdbFile(os.path.join(ENV, uuid, "data.mdb"))
    main_db = dict(env.items("main"))
    out = {"source": f"crates/meilisearch/tests/upgrade/v1_12/v1_12_0.ms/indexes/{uuid}/data.mdb", "fst": [], "bitmaps": []}
    word_keys = [k.decode() for k, _ in env.items("word-docids")]
    for name in ("words-fst", "stop-words", "exact-words", "words-prefixes-fst"):
        out["fst"].append({"name": f"main[{name}]", "hex": main_db[name.encode()].hex(),
                           "contains": word_keys if name == "words-fst" else None, "keys": None})
    facet_values = {}
    for k, _ in env.items("facet-id-normalized-string-strings"):
        facet_values.setdefault(k[:2].hex(), []).append(k[2:].decode())
    for k, v in env.items("facet-id-string-fst"):
        out["fst"].append({"name": f"facet-id-string-fst[{k.hex()}]", "hex": v.hex(), "contains": None,
                           "keys": facet_values[k.hex()]})
    # posting lists as milli wrote them (CboRoaringBitmapCodec: <= 7 docids raw, else RoaringBitmap) + one RoaringBitmap
    out["bitmaps"].append({"name": "main[documents-ids]", "codec": "roaring", "hex": main_db[b"documents-ids"].hex(),
                           "n_documents": sum(1 for _ in env.items("documents"))})
    for db in ("word-docids", "exact-word-docids", "word-pair-proximity-docids", "word-position-docids"):
        for k, v in env.items(db):
            out["bitmaps"].append({"name": f"{db}[{k.hex()}]", "codec": "cbo", "hex": v.hex()})
    # the inverted-index databases themselves (2 documents), to hold the toy indexer of tests/toy_milli.py against
    # what milli's write path produced.  Documents as (fid -> JSON) from the obkv records; charabia already
    # normalised the words in the databases ("très" -> "tres"), the test feeds the toy the normalised text.
    fields = json.loads(main_db[b"fields-ids-map"])["ids_names"]
    docs = []
    for _, v in env.items("documents"):
        doc, i = {}, 0
        while i < len(v):
            fid = struct.unpack_from(">H", v, i)[0]
            i += 2
            ln, shift = 0, 0
            while True:                       # obkv: LEB128 value length
                b = v[i]
                i += 1
                ln |= (b & 0x7F) << shift
                shift += 7
                if not b & 0x80:
                    break
            doc[fields[str(fid)]] = json.loads(v[i:i + ln])
            i += ln
        docs.append(doc)

    def ids(v):
        return list(struct.unpack("<%dI" % (len(v) // 4), v))
    dbs = {"word_docids": [[k.decode(), ids(v)] for k, v in env.items("word-docids")],
           "exact_word_docids": [[k.decode(), ids(v)] for k, v in env.items("exact-word-docids")],
           "word_fid_docids": [[k[:-3].decode(), struct.unpack(">H", k[-2:])[0], ids(v)]
                               for k, v in env.items("word-field-id-docids")],
           "word_position_docids": [[k[:-3].decode(), struct.unpack(">H", k[-2:])[0], ids(v)]
                                    for k, v in env.items("word-position-docids")],
           "field_id_word_count_docids": [[struct.unpack(">H", k[:2])[0], k[2], ids(v)]
                                          for k, v in env.items("field-id-word-count-docids")],
           "word_pair_proximity_docids": [[k[0], k[1:].split(b"\0")[0].decode(), k[1:].split(b"\0")[1].decode(), ids(v)]
                                          for k, v in env.items("word-pair-proximity-docids")]}
    out["index"] = {"documents": docs, "fields": fields, "exact_attributes": ["surname"],
                    "stop_words": [k.decode() for k in __import__("oracle.fst_oracle", fromlist=["x"]).fst_keys(main_db[b"stop-words"])],
                    "databases": dbs}
    # arroy item leaves of the one embedder (key = u16 index, mode byte, u32 item; value = tag 0, f32 norm, f32 x d):
    # the norm arroy stored at indexing time pins the oracle's f32 norm on real 384-d embeddings
    out["arroy_items"] = [{"key": k.hex(), "docid": struct.unpack(">I", k[3:7])[0], "norm_f32_hex": v[1:5].hex(),
                           "vector_f32_hex": v[5:].hex()}
                          for k, v in env.items("vector-arroy") if len(v) == 1 + 4 + 4 * 384 and v[0] == 0]
    json.dump(out, open(OUT, "w"), indent=0, sort_keys=True, ensure_ascii=False)
    print(len(out["fst"]), "fst blobs,", len(out["bitmaps"]), "bitmaps ->", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
