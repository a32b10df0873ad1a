"""Derives profiles/pmc_lauum.json and the tables of profiles/README.md from the rocprofv3 CSVs that
tools/collect_profiles.sh produced (run on the dev box after copying them into profiles/)."""
import csv, collections, json, os, re, sys
TAG = sys.argv[1] if len(sys.argv) > 1 else "r04"
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
P = os.path.join(ROOT, "profiles")
N = 16384

def per_kernel(path):
    d = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        d.setdefault((r["Dispatch_Id"], r["Kernel_Name"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    return d

def avg(d, key, pred):
    v = [c[key] for (_, k), c in d.items() if pred(k)]
    return sum(v) / len(v), len(v)

f = per_kernel(os.path.join(P, TAG + "_pmc_FETCH_SIZE_fit_n16384.csv"))
w = per_kernel(os.path.join(P, TAG + "_pmc_WRITE_SIZE_fit_n16384.csv"))
m = per_kernel(os.path.join(P, TAG + "_pmc_MFMA_fit_n16384.csv"))
stats = {r["Name"]: r for r in csv.DictReader(open(os.path.join(P, TAG + "_bench_kernel_stats.csv")))}
line = json.load(open(os.path.join(P, TAG + "_bench_line.json")))

lau = lambda k: "true, true" in k
fl, n = avg(f, "FETCH_SIZE", lau)
wl, _ = avg(w, "WRITE_SIZE", lau)
tr, _ = avg(f, "FETCH_SIZE", lambda k: "trmv_lower" in k)
lau_ms = float(stats["void gemm_tiles_kernel<true, true, 0, 4, 128, 128>(GemmArgs)"]["AverageNs"]) / 1e6
out = {
    "kernel": "gemm_tiles_kernel<true, true, 0, 4, 128, 128> (K^-1 = L^-T L^-1)", "N": N, "launches_sampled": n,
    "FETCH_SIZE_KB_raw": fl, "WRITE_SIZE_KB_raw": wl,
    "calibration": {"trmv_lower_kernel_FETCH_SIZE_KB": tr, "its_true_read_KB": N * N / 2 * 8 / 1024,
                    "ratio": tr / (N * N / 2 * 8 / 1024),
                    "note": "wide 16-byte coalesced reads: FETCH_SIZE = 0.50 x true bytes on gfx950 (MI355X_MICROARCH.md, "
                            "HBM section) -> doubled below; WRITE_SIZE matches kmat's known 1056768 KB exactly -> used as is"},
    "hbm_bytes_per_launch": (2 * fl + wl) * 1024,
    "algorithmic_flop_per_launch": N ** 3 / 3,
    "compulsory_bytes_per_launch": float(N * N / 2 * 8 * 2),
    "avg_launch_ms_rocprof_stats": lau_ms,
    "comment": "FETCH counts L2 misses (incl. Infinity-Cache hits): a 128x128-tile product re-reads its operand panels once "
               "per tile (91.6 GB requested by construction); L2 absorbs about a third.  The launch runs at the fp64-MFMA "
               "issue rate and takes the same time when every operand row aliases row 0 (tools/lauum_probe.hip): it is not "
               "limited by this traffic."}
json.dump(out, open(os.path.join(P, "pmc_lauum.json"), "w"), indent=1)

fam = collections.OrderedDict()
for (_, k), c in m.items():
    a = fam.setdefault(k.split("(")[0].replace("void ", ""), collections.Counter())
    for kk, v in c.items():
        a[kk] += v
    a["n"] += 1
rows = ["| kernel | launches | MFMA-busy / busy-CU SIMD cycles | MFMA-busy / all SIMD cycles of the launch window |", "|---|---|---|---|"]
for k, a in fam.items():
    if a["SQ_INSTS_VALU_MFMA_MOPS_F64"] > 0:
        rows.append("| `%s` | %d | %.3f | %.3f |" % (k, a["n"], a["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * a["SQ_BUSY_CU_CYCLES"]),
                                                     a["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * 256 * a["GRBM_GUI_ACTIVE"] / 8)))
mfma_table = "\n".join(rows)

def hbm_row(label, pred, statname):
    fk, _ = avg(f, "FETCH_SIZE", pred)
    wk, _ = avg(w, "WRITE_SIZE", pred)
    ns = float([r for nme, r in stats.items() if statname in nme][0]["AverageNs"])
    byts = (2 * fk + wk) * 1024
    return "| %s | %.2f GB | %.0f us | %.1f TB/s |" % (label, byts / 1e9, ns / 1e3, byts / ns / 1e3)
hbm = ["| kernel | bytes moved (FETCH x2 + WRITE) | duration | rate (peak ~ 8 TB/s) |", "|---|---|---|---|",
       hbm_row("`trmv_lower_kernel` (z = L^-1 y)", lambda k: "trmv_lower" in k, "trmv_lower_kernel"),
       hbm_row("`kmat_kernel<1, double>` (lower tiles of K; one `kf_sqrt` + `kf_exp_neg` per entry, round 6)", lambda k: "kmat_kernel" in k, "kmat_kernel<1"),
       hbm_row("`gemv_t_tri_kernel` (alpha = L^-T z; round 6: row chunks of ~np/32 rows, equal work per workgroup)", lambda k: "gemv_t_tri" in k, "gemv_t_tri_kernel"),
       hbm_row("`grad_reduce_kernel<1, double>` (K^-1 . dK/dtheta sums; kernel derivative recomputed per entry; ALU-bound: ~65 fp64 instructions per entry)", lambda k: "grad_reduce" in k, "grad_reduce_kernel<1")]

readme = open(os.path.join(P, "README.md")).read()
readme = re.sub(r"<!-- MFMA_TABLE -->.*?<!-- /MFMA_TABLE -->", "<!-- MFMA_TABLE -->\n" + mfma_table + "\n<!-- /MFMA_TABLE -->", readme, flags=re.S)
readme = re.sub(r"<!-- HBM_TABLE -->.*?<!-- /HBM_TABLE -->", "<!-- HBM_TABLE -->\n" + "\n".join(hbm) + "\n<!-- /HBM_TABLE -->", readme, flags=re.S)
readme = re.sub(r"<!-- LAUUM -->.*?<!-- /LAUUM -->", "<!-- LAUUM -->`AverageNs` in the stats CSV: %.2f ms; `roofline.avg_launch_ms` of the committed bench line: %.2f ms (%.1f TFLOP/s, frac %.3f); L2-miss traffic %.1f GB per launch<!-- /LAUUM -->" % (
    lau_ms, line["roofline"]["dominant_launch"]["avg_launch_ms"], line["roofline"]["dominant_launch"]["achieved"], line["roofline"]["dominant_launch"]["achieved"] / 78.6, out["hbm_bytes_per_launch"] / 1e9), readme, flags=re.S)
open(os.path.join(P, "README.md"), "w").write(readme)
print(mfma_table); print("\n".join(hbm)); print(json.dumps({k: out[k] for k in ("hbm_bytes_per_launch", "avg_launch_ms_rocprof_stats")}))
