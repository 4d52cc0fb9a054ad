#!/usr/bin/env python
"""Symbolises a RB_PROFILE dump of tools/ranked_bench (raw return addresses + the process's maps): inclusive and leaf
sample counts per function.  python tools/r3_symbolize.py <dump> [top]"""
import collections
import os
import subprocess
import sys

path = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 45
SKIP = 0 if os.environ.get("ALLOC_SITES") else 2   # tools/alloc_sites.cpp dumps start at the caller of operator new
maps, samples, sizes = [], [], []
for line in open(path):
    if line.startswith("M "):
        f = line[2:].split()
        lo, hi = (int(x, 16) for x in f[0].split("-"))
        off = int(f[2], 16)
        name = f[5] if len(f) > 5 else ""
        maps.append((lo, hi, off, name))
    elif line.startswith("S"):
        samples.append([int(x, 16) for x in line.split()[1:]])
    elif line.startswith("Z "):
        sizes.append(int(line.split()[1]))
# module base = lowest mapping of each file
base = {}
for lo, hi, off, name in maps:
    if name and (name not in base or lo - off < base[name]):
        base[name] = lo - off
def locate(pc):
    for lo, hi, off, name in maps:
        if lo <= pc < hi:
            return name, pc - base[name]
    return None, pc
by_mod = collections.defaultdict(set)
for s in samples:
    for pc in s:
        mod, rel = locate(pc - 1)
        if mod:
            by_mod[mod].add(rel)
sym = {}
SYMB = "/opt/rocm/lib/llvm/bin/llvm-symbolizer"
for mod, rels in by_mod.items():
    local = mod
    if not os.path.exists(local):
        # the GPU box's scratch path -> this checkout
        for marker in ("/meilisearch_amd/", "/tools/bin/", "/oracle/"):
            if marker in mod:
                local = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), marker.strip("/"), mod.split(marker, 1)[1])
    if not os.path.exists(local):
        for r in rels:
            sym[(mod, r)] = os.path.basename(mod)
        continue
    # nm-based: the defined text symbols of the module, bisected (llvm-symbolizer answers ?? for HIP fat binaries)
    import bisect
    out = subprocess.run(["nm", "-C", "--defined-only", local], capture_output=True, text=True).stdout
    if not out.strip():
        out = subprocess.run(["nm", "-C", "-D", "--defined-only", local], capture_output=True, text=True).stdout
    table = []
    for line in out.splitlines():
        f = line.split(None, 2)
        if len(f) == 3 and f[1] in "tTwW":
            table.append((int(f[0], 16), f[2]))
    table.sort()
    addrs = [a for a, _ in table]
    for r in rels:
        i = bisect.bisect_right(addrs, r) - 1
        sym[(mod, r)] = table[i][1] if i >= 0 else os.path.basename(mod)
# EXCLUDE="a,b": drop the samples with a frame whose name contains one of these (an emulated run's kernel emulation:
# EXCLUDE=hipemu::,pthread_sigmask leaves the host logic of the search threads)
excl = [x for x in os.environ.get("EXCLUDE", "").split(",") if x]
if excl:
    kept = []
    for s in samples:
        names = [sym.get(locate(pc - 1), "?") for pc in s[SKIP:]]
        if not any(x in nm for nm in names for x in excl):
            kept.append(s)
    print("EXCLUDE: kept", len(kept), "of", len(samples), "samples")
    samples = kept
incl, leaf = collections.Counter(), collections.Counter()
for s in samples:
    names = []
    for pc in s[SKIP:]:   # skip the handler and the signal trampoline
        mod, rel = locate(pc - 1)
        names.append(sym.get((mod, rel), "?") if mod else "?")
    if not names:
        continue
    leaf[names[0]] += 1
    for n in set(names):
        incl[n] += 1
n = len(samples)
print(n, "samples")
print("---- leaf")
for k, v in leaf.most_common(top):
    print("%6.2f%%  %s" % (100.0 * v / n, k[:150]))
print("---- inclusive")
for k, v in incl.most_common(top):
    print("%6.2f%%  %s" % (100.0 * v / n, k[:150]))

# ---- who calls the allocator: for samples whose leaf is in malloc / free / operator new (or libc's unnamed internals next
# to them), the first frame above that belongs to libmsi / the driver
alloc_leaf = ("malloc", "free", "operator new", "operator delete", "__default_morecore", "__lll_lock", "realloc", "cfree", "_M_fill_insert")
callers = collections.Counter()
n_alloc = 0
for s in samples:
    names = []
    for pc in s[SKIP:]:
        mod, rel = locate(pc - 1)
        names.append((sym.get((mod, rel), "?"), os.path.basename(mod or "?")))
    if not names or not any(names[0][0].startswith(a) for a in alloc_leaf):
        continue
    n_alloc += 1
    for fn, mod in names[1:]:
        if mod.startswith("libmsi") or mod.startswith("ranked_bench"):
            callers[fn[:110]] += 1
            break
print("---- allocator samples: %d of %d (%.1f%%); first libmsi / driver frame above them" % (n_alloc, n, 100.0 * n_alloc / max(1, n)))
for k, v in callers.most_common(25):
    print("%6.2f%%  %s" % (100.0 * v / n, k))

if sizes:   # tools/alloc_sites.cpp: bytes by allocating function (leaf) and the large requests
    by_bytes, big = collections.Counter(), collections.Counter()
    for s, z in zip(samples, sizes):
        mod, rel = locate(s[0] - 1)
        name = sym.get((mod, rel), "?") if mod else "?"
        by_bytes[name[:110]] += z
        if z >= 16384:
            big[(name[:90], z)] += 1
    tot = sum(sizes)
    print("---- sampled bytes by allocating function (%d bytes in %d samples)" % (tot, len(sizes)))
    for k, v in by_bytes.most_common(15):
        print("%6.2f%%  %s" % (100.0 * v / tot, k))
    print("---- requests of 16 KiB and more: (function, size) x samples")
    for (k, z), v in big.most_common(15):
        print("%5d x %8d  %s" % (v, z, k))

if os.environ.get("ALLOC_SITES_CALLERS"):   # who calls the std:: internals that lead the leaf list (frame above the leaf)
    up = collections.Counter()
    for s in samples:
        names = []
        for pc in s[SKIP:SKIP + 4]:
            mod, rel = locate(pc - 1)
            names.append(sym.get((mod, rel), "?")[:70] if mod else "?")
        if names and names[0].startswith(("std::", "void std::")):
            up[" <- ".join(names[:3])] += 1
    print("---- callers of std:: leaves")
    for k, v in up.most_common(20):
        print("%6.2f%%  %s" % (100.0 * v / n, k))
