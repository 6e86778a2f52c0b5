#!/usr/bin/env python3
"""profiles/rNN_traffic_b<chunk>.json from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace):

    make_traffic.py fetch_counter_collection.csv write_counter_collection.csv batch out.json

Per-launch means in KB as reported by rocprofv3, keyed by bench.py's launch names.  Kernels that serve several
layers are told apart by grid size (largest grid = the high-resolution layer)."""
import collections, csv, json, sys

# launch name -> (kernel name prefix, rank of the grid size among that kernel's launches, 0 = largest)
LAUNCHES = {
    "conv3x3_det": ("void hfnet::k_conv3x3_wlds<4, false", 0),
    "conv3x3_desc_taps": ("void hfnet::k_conv3x3_wlds<4, true", 0),
    "stem_block_L02": ("void hfnet::k_stem_block2<24, 16>", 0),
    "block_L03": ("void hfnet::k_block_fused8<2, 1, 2, false", 0),
    "block_L04": ("void hfnet::k_block_fused4<1, 1, 3, true", 0),
    "block_L05": ("void hfnet::k_block_fused8<2, 1, 3, false", 0),
    "block_L06": ("void hfnet::k_block_fused6<6, 3, false", 0),
    "block_L07": ("void hfnet::k_block_fused8<1, 3, 6, false", 0),
    "block_L08": ("void hfnet::k_block_fused8<2, 2, 12, false, 1>", 0),
    "pointwise_desc_taps": ("void hfnet::k_pointwise_wlds<4>", 0),
    "det_tail": ("hfnet::k_det_tail", 0),
    "fc": ("void hfnet::k_fc_mfma<16>", 0),
    "block_L09": ("void hfnet::k_block_fused6<12, 3, true", 0),
    "block_L12": ("void hfnet::k_block_fused6<12, 5, false", 0),
    "block_L13": ("void hfnet::k_block_fused6<18, 5, true", 0),
    "nms_mask": ("hfnet::k_nms_mask", 0),
    "nms_select": ("hfnet::k_nms_select", 0),
    "sample": ("hfnet::k_sample", 0),
    "pyramid_resize": ("hfnet::k_resize_u8", 0),
    "depthwise_L16": ("void hfnet::k_depthwise<1, 5>", 0),
    "vlad_aggregate": ("void hfnet::k_vlad_aggregate<16>", 0),
    "match_gemm": ("hfnet::k_bow_gemm_cand", 0),
    "match_candidates": ("hfnet::k_bow_candidates", 0),
}


def per_launch(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))     # kernel -> grid -> values
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"]][int(r["Grid_Size"])].append(float(r["Counter_Value"]))
    return acc


def pick(acc, prefix, rank):
    for k, grids in acc.items():
        if k.startswith(prefix):
            g = sorted(grids, reverse=True)[rank]
            v = grids[g]
            return k, sum(v) / len(v)
    return None, None


fetch, write = per_launch(sys.argv[1]), per_launch(sys.argv[2])
out = {"_provenance": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace) around `python bench.py --steps 1 --warmup 1 "
                      "--chunk %s --configs none --no-cpu-baseline` on MI355X; per-launch means in KB as reported; bench.py applies MI355X_MICROARCH.md's gfx950 "
                      "correction: HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.  Made by tools/make_traffic.py." % sys.argv[3],
       "batch": int(sys.argv[3]), "kernels": {}}
for name, (prefix, rank) in LAUNCHES.items():
    k, f = pick(fetch, prefix, rank)
    _, w = pick(write, prefix, rank)
    if k is not None and w is not None:
        out["kernels"][name] = {"kernel": k.split("(")[0], "fetch_kb": f, "write_kb": w}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
