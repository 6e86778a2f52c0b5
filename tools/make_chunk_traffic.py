#!/usr/bin/env python3
"""HBM-side bytes of ONE extract + match call (chunk) from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace) of a
short bench.py run:

    make_chunk_traffic.py fetch_counter_collection.csv write_counter_collection.csv batch out.json [note]

Sums the counters over every hfnet:: kernel of the run and divides by the number of calls (= dispatches of the stem kernel, one per call).
bench.py reads `chunk_hbm_bytes` (= (2 x FETCH_SIZE + WRITE_SIZE) x 1024, the gfx950 correction of MI355X_MICROARCH.md) for roofline_bf16x3."""
import collections, csv, json, sys


def fold(path):
    tot = collections.defaultdict(float)
    calls = collections.defaultdict(int)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if "hfnet::" not in k:
            continue
        tot[k] += float(r["Counter_Value"])
        calls[k] += 1
    return tot, calls


f, fc = fold(sys.argv[1])
w, wc = fold(sys.argv[2])
stem = [k for k in fc if "k_stem_block2" in k]
n_calls = fc[stem[0]] if stem else 0
if not n_calls:
    sys.exit("no stem kernel dispatches found: cannot count the calls")
fk, wk = sum(f.values()) / n_calls, sum(w.values()) / n_calls
per = sorted(((k, f[k] / n_calls, w.get(k, 0.0) / n_calls) for k in f), key=lambda t: -(2 * t[1] + t[2]))
out = {"_provenance": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace) around a short bench.py run; counters summed over every "
                      "hfnet:: kernel, divided by the %d calls of the run (stem kernel dispatches); KB as reported; chunk_hbm_bytes = (2*FETCH + WRITE) * 1024 "
                      "(gfx950: FETCH_SIZE counts 128-byte requests as 64).  Made by tools/make_chunk_traffic.py.%s" % (n_calls, (" " + sys.argv[5]) if len(sys.argv) > 5 else ""),
       "batch": int(sys.argv[3]), "calls": n_calls, "fetch_kb_per_chunk": fk, "write_kb_per_chunk": wk, "chunk_hbm_bytes": (2 * fk + wk) * 1024.0,
       "largest": [{"kernel": k[:90], "fetch_kb": a, "write_kb": b, "hbm_gb": (2 * a + b) * 1024 / 1e9} for k, a, b in per[:16]]}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print("calls %d  fetch %.1f MB  write %.1f MB  hbm-side %.2f GB per call" % (n_calls, fk / 1e3, wk / 1e3, out["chunk_hbm_bytes"] / 1e9))
