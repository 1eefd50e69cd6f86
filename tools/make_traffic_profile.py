"""profiles/r04_traffic.json: per roofline kernel of bench.py the HBM bytes per launch and the matrix-pipe busy cycles measured
by rocprofv3 --pmc INSIDE the kernel's workload (tools/calls/r04_pmc.sh: separate FETCH_SIZE / WRITE_SIZE / SQ passes over the
eager window step; tools/pmc_summarize.py averages per kernel).  gfx950: FETCH_SIZE counts wide coalesced reads at half their
bytes (MI355X_MICROARCH.md, HBM) -> hbm bytes = 2 x FETCH_SIZE + WRITE_SIZE.  bench.py reads this file for `roofline.traffic` /
`roofline.mfma_util`; the timings in the bench line are measured live."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
src = {"mcat": "r04_pmc_mcat.json", "nacagat": "r04_pmc_nacagat.json", "f32": "r04_pmc_mcat_f32_100k.json"}
want = {   # bench kernel key -> (source, substring of the summarised kernel name, workload, algorithmic bytes)
    "patch_fc_fwd_kernel<1024->256, bf16>": ("mcat", "patch_fc_fwd_kernel", "MCAT medium, 32 x 15000 x 1024 bf16 window", 32 * 15000 * 1280 * 2),
    "coattn_fwd_partial_kernel<256,bf16>": ("mcat", "coattn_fwd_partial_kernel<256, false>", "MCAT medium, 32 x 15000 x 256 bf16 H_bag", 32 * 15000 * 256 * 2),
    "bag_rowdot_gated_exact_kernel<256> (f32 key bag)": ("nacagat", "bag_rowdot_gated_exact_kernel", "NaCAGaT medium, 32 x 15000 x 256 fp32 key bag", 32 * 15000 * 256 * 4),
    "coattn_fwd_partial_kernel<256,f32>": ("f32", "coattn_fwd_partial_kernel<256, true>", "MCAT medium, 8 x 100000 x 256 fp32 H_bag", 8 * 100000 * 256 * 4),
}
out = {"_source": "tools/calls/r04_pmc.sh -> profiles/r04_pmc_*.json (rocprofv3 --pmc, separate passes, kernels inside the eager window step)",
       "_correction": "hbm_bytes_per_launch = 2 * FETCH_SIZE_KB * 1024 + WRITE_SIZE_KB * 1024 (gfx950 FETCH_SIZE undercount of wide reads)"}
for key, (s, sub, workload, alg) in want.items():
    d = json.load(open(os.path.join(P, src[s])))
    name = next(k for k in d if sub in k)
    e = d[name]
    out[key] = {"workload": workload, "pmc_kernel": name, "hbm_bytes_per_launch": e.get("hbm_bytes_per_launch"),
                "algorithmic_bytes_per_launch": alg, "SQ_VALU_MFMA_BUSY_CYCLES": e.get("SQ_VALU_MFMA_BUSY_CYCLES"),
                "lds_bank_conflict_share": e.get("lds_bank_conflict_share"), "avg_us_in_pmc_passes": e.get("avg_us_by_pass"),
                "pmc_file": "profiles/" + src[s]}
json.dump(out, open(os.path.join(P, "r04_traffic.json"), "w"), indent=1)
for k, v in out.items():
    if not k.startswith("_"):
        print(k, v["hbm_bytes_per_launch"], v["algorithmic_bytes_per_launch"], round(v["hbm_bytes_per_launch"] / v["algorithmic_bytes_per_launch"], 3), v["SQ_VALU_MFMA_BUSY_CYCLES"])
