"""HBM bytes per launch of the hot kernels from the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_passes.sh, stamped with
the hash of the library sources they were measured on (bench.py quotes the figures only for that build).
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KB and FETCH_SIZE reports half the bytes of wide
coalesced reads on gfx950 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is uncalibrated there -- ratios between builds
are exact, absolutes are +-.     python tools/pmc_traffic.py <pmc outdir> <out.json>"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

STAGE = {"seq_fwdh_kernel": "seq_fwd", "seq_bwdh_kernel": "seq_bwd", "wgradh_kernel": "wgrad", "seq_fwd3_kernel": "seq_fwd", "seq_bwd3_kernel": "seq_bwd", "wgrad3_kernel": "wgrad", "wgrad4_kernel": "wgrad", "gather_kernel": "gather",
         "merw_walk_kernel": "sampler_walk", "merw_walk_otf_kernel": "sampler_walk"}


def per_kernel(root):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(root + "/pass*/**/*counter_collection.csv", recursive=True)):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
                name = name.split("(")[0].split("<")[0].strip()
                acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out, l2 = {}, {}
    for k, cs in acc.items():
        if k in STAGE and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            f = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"])
            w = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
            out[STAGE[k]] = (2.0 * f + w) * 1024.0
        if k in STAGE and "TCC_REQ_sum" in cs:          # L2 <- CU requests of 128 B (bench.py: roofline.bound_evidence.l2_frac)
            l2[STAGE[k]] = sum(cs["TCC_REQ_sum"]) / len(cs["TCC_REQ_sum"])
    return out, l2


def main():
    import bench
    root, dst = sys.argv[1], sys.argv[2]
    res = {"source_hash": bench.source_hash(),
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/pmc_passes.sh) on this build; bytes = "
                     "(2*FETCH_SIZE + WRITE_SIZE) * 1024 per launch, FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section",
           }
    res["hbm_bytes_per_launch"], res["l2_requests_per_launch"] = per_kernel(os.path.join(root, "cora"))
    if os.path.isdir(os.path.join(root, "pubmed")):
        res["pubmed_hbm_bytes_per_launch"], res["pubmed_l2_requests_per_launch"] = per_kernel(os.path.join(root, "pubmed"))
    json.dump(res, open(dst, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
