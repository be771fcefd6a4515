"""HBM traffic per launch from the two PMC passes of tools/profile_round.sh (FETCH_SIZE, WRITE_SIZE),
per kernel group of the BASELINE configs[1] step -> profiles/pmc_traffic.json (read by bench.py for
roofline.traffic).

    python tools/pmc_traffic.py gpurun_out/prof_<tag> <tag>

Corrections (MI355X_MICROARCH.md, HBM section): rocprofv3 reports both counters in KiB; on gfx950
FETCH_SIZE tallies a 128-B request of a wide (16 B/lane) coalesced read as 64 B -> doubled here.  Both are
calibrated inside the same run on kernels with a known byte count: the config-4 gather reads
2^22 x (256 B row + 8 B id) and FETCH_SIZE reports 0.5007 of it; k_fill_table writes the 256 GB table and
WRITE_SIZE reports 250 000 000 KiB (exact)."""
import collections
import csv
import json
import sys

root, tag = sys.argv[1], sys.argv[2]
# (kernel name prefix, grid size) -> group, for F=26 D=16 X=13 FC[512,256,1] B=4096 (64x64 tiles, 256 threads)
GROUPS = {
    ("k_gemm_tn", 57344): "fc_bwd_dw0",       # 7 x 8 tiles x 4 batch splits (4-wave workgroups, rounds 1-2)
    ("k_gemm_tn", 64512): "fc_bwd_dw1",       # 9 x 4 tiles x 7 batch splits
    ("k_gemm_tn", 114688): "fc_bwd_dw0",      # ... 8-wave workgroups (round 3: two wave groups split every slab)
    ("k_gemm_tn", 129024): "fc_bwd_dw1",
    ("k_gemm_nt", 114688): "fc_bwd_data0",    # 64 x 7 tiles
    ("k_gemm_nt", 65536): "fc_fwd1",          # 64 x 4 tiles
    ("k_gemm_nt", 131072): "fc_fwd0|fc_bwd_data1",   # 64 x 8 tiles each (same grid: averaged)
    ("k_fc_fwd_pair", None): "fc_fwd01",      # round 3: the first two forward GEMMs in one launch
    ("k_emb_fwd", 159744): "emb_fwd",
    ("k_emb_fwd", 16777216): "gather_c4_single_hot",
    ("k_emb_fwd", 2097152): "gather_c4_bags32",
    ("k_emb_reduce_update", None): "emb_bwd_update",
    ("k_last_bwd", None): "head_last_bwd",
    ("k_dense_update", None): "dense_update",
    ("k_dense_prereduce", None): "dense_prereduce",
    ("k_radix_scatter", None): "emb_sort_scatter",
    ("k_field_sort_segments", None): "emb_sort",
    ("k_fill_table", None): "fill_table_256GB",
}
acc = {}
for which in ("fetch", "write"):
    a = collections.defaultdict(list)
    for r in csv.DictReader(open("%s/pmc_%s/c2_counter_collection.csv" % (root, which))):
        nm = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        base = nm.split("<")[0].split("(")[0]
        g = GROUPS.get((base, int(r["Grid_Size"]))) or GROUPS.get((base, None))
        if g:
            a[g].append(float(r["Counter_Value"]))
    for g, v in a.items():
        acc.setdefault(g, {})[which] = (len(v), sum(v) / len(v))
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), %s" % tag,
       "corrections": "KiB -> bytes; FETCH_SIZE x2 (gfx950 tallies 128-B requests of wide coalesced reads at 64 B)", "groups": {}}
for g, v in sorted(acc.items()):
    f, w = v.get("fetch", (0, 0.0)), v.get("write", (0, 0.0))
    rec = {"dispatches": f[0], "fetch_size_KiB_raw": round(f[1], 1), "write_size_KiB_raw": round(w[1], 1),
           "read_bytes": 2.0 * f[1] * 1024, "write_bytes": w[1] * 1024}
    rec["hbm_bytes_per_launch"] = rec["read_bytes"] + rec["write_bytes"]
    for name in g.split("|"):
        out["groups"][name] = rec
try:        # (keys other tools keep in the same file: tools/adam_profile.py's "fused_adam_hbm")
    old = json.load(open("profiles/pmc_traffic.json"))
    for k, v in old.items():
        out.setdefault(k, v)
except (OSError, ValueError):
    pass
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
for g, r in out["groups"].items():
    print("%-24s read %8.2f MB  write %8.2f MB" % (g, r["read_bytes"] / 1e6, r["write_bytes"] / 1e6))
