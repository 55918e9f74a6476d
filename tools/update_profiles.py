"""copy the judged summaries of gpurun_out/prof_<tag>/ (tools/profile_bench.sh) into profiles/ and refresh the entry of
profiles/pmc_traffic.json for one bench config.   usage: python tools/update_profiles.py <tag> <config> [fetch_factor]
fetch_factor: HBM bytes / (FETCH_SIZE * 1024) from tools/pmc_calibrate.sh -> profiles/<round>_pmc_calibration.json, which must
exist (the entry cites it): 2.0 on every read pattern of these kernels -- 4-B and 16-B per lane streaming, 16-B records gathered
by sorted slot lists of any density and by a permutation (the L2 fetches 128-byte lines, the counter tallies 64 per request)."""
import csv, glob, json, os, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from bench import kernel_source_hash  # noqa: E402
tag = sys.argv[1]
config = sys.argv[2]
fetch_factor = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
write_factor = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
calibration = f"profiles/{tag.split('_')[0]}_pmc_calibration.json"
if not os.path.exists(os.path.join(root, calibration)):
    sys.exit(f"{calibration} does not exist: run tools/pmc_calibrate.sh on the GPU box and copy gpurun_out/pmc_calibrate/calibration.json there first")
shutil.copy(os.path.join(src, "summary.txt"), os.path.join(dst, f"{tag}_rocprofv3_summary.txt"))
shutil.copy(os.path.join(src, "summary.json"), os.path.join(dst, f"{tag}_rocprofv3_summary.json"))
ks = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
if ks:
    shutil.copy(max(ks, key=os.path.getmtime), os.path.join(dst, f"{tag}_kernel_stats.csv"))
b = os.path.join(src, "bench_under_rocprof.json")
if os.path.exists(b):
    shutil.copy(b, os.path.join(dst, f"{tag}_bench_under_rocprof.json"))
kernel = "k_us_bounce" if config.startswith("us_") else "k_bounce"
# BVH scenes: a bounce is k_trace + k_shade (kernels_wavefront.h), k_trace + k_us_shade in ultrasound mode (kernels_us_wavefront.h)
families = (kernel, kernel + "_pool")
if config == "testring":
    families = ("k_trace_primary", "k_trace", "k_shade")
    kernel = "k_trace_primary + k_trace + k_shade"
if config == "usmain_loop":   # the reference's us_render loop: the roofline kernel of that line is the beamformer
    families = ("k_das_beamform",)
    kernel = "k_das_beamform"
if config == "us_testring":
    families = ("k_trace", "k_us_shade")
    kernel = "k_trace + k_us_shade"


def per_dispatch(sub, ctr):
    f = max(glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)  # newest run
    tot, n = {}, {}
    rows = [r for r in csv.DictReader(open(f))
            if r["Counter_Name"] == ctr and any(k + "<" in r["Kernel_Name"] for k in families)]
    # the first render of a brute-force scene starts with a 2-spp probe pass (the library learns its launch plan from it): those
    # small launches are not the workload -- keep the dispatches of the full-size grid only
    fam = lambda r: r["Kernel_Name"].split("<")[0]   # (k_trace and k_shade have grids of their own)
    gmax = {}
    for r in rows:
        gmax[fam(r)] = max(gmax.get(fam(r), 0), int(r["Grid_Size"]))
    for r in rows:
        if int(r["Grid_Size"]) * 2 < gmax[fam(r)]:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        tot[k] = tot.get(k, 0.0) + float(r["Counter_Value"])
        n[k] = n.get(k, 0) + 1
    return {k: tot[k] / n[k] for k in tot}, n


fetch, nf = per_dispatch("pmc_fetch", "FETCH_SIZE")
write, nw = per_dispatch("pmc_write", "WRITE_SIZE")
num = sum((fetch_factor * fetch[k] + write_factor * write.get(k, 0.0)) * 1024 * nf[k] for k in fetch)
path = os.path.join(dst, "pmc_traffic.json")
allrec = json.load(open(path)) if os.path.exists(path) else {}
if "k_bounce_hbm_bytes_per_launch" in allrec:   # round-1 layout: keep it under its own key
    allrec = {"r01_cbox": allrec}
# VALU-issue utilisation of the dominant kernel family (4th pass of profile_bench.sh): a wave64 VALU instruction occupies its SIMD-32
# for 2 cycles, SQ_BUSY_CYCLES is summed over the 32 shader engines of the 1024 SIMDs -> busy = SQ_INSTS_VALU / (16 * SQ_BUSY_CYCLES)
valu_busy = lane_active = None
try:
    iv, ni = per_dispatch("pmc_sq", "SQ_INSTS_VALU")
    bc, nb = per_dispatch("pmc_sq", "SQ_BUSY_CYCLES")
    tot_c = sum(bc[k] * nb[k] for k in bc)
    valu_busy = round(sum(iv[k] * ni[k] for k in iv) / (16.0 * tot_c), 4) if tot_c else None
    # share of the lanes active per VALU instruction (idle lanes of a chain launch, partial waves, divergence)
    tc, nt = per_dispatch("pmc_sq", "SQ_THREAD_CYCLES_VALU")
    ai, na = per_dispatch("pmc_sq", "SQ_ACTIVE_INST_VALU")
    tot_a = sum(ai[k] * na[k] for k in ai)
    lane_active = round(sum(tc[k] * nt[k] for k in tc) / (64.0 * tot_a), 4) if tot_a else None
except (ValueError, KeyError):
    pass
# per kernel family: PMC bytes per bench step against the byte model of the same step (bench_under_rocprof.json carries the
# model's split for BVH scenes); dispatches per step = dispatches of the two profiled steps / 2
by_kernel = None
try:
    bj = json.load(open(b))
    alg = bj["roofline"].get("algorithmic_bytes_per_step")
    if alg and config.startswith("us_") and "k_us_shade" not in alg:   # records made before the ultrasound families had their names
        alg = {"k_trace": alg["k_trace_primary + k_trace"], "k_us_shade": alg["k_shade"]}
    if alg:
        steps_pmc = 2.0
        fam_bytes = {}
        for k in fetch:
            walk, shade = ("k_trace", "k_us_shade") if config.startswith("us_") else ("k_trace_primary + k_trace", "k_shade")
            fam = shade if k.startswith(("k_shade", "k_us_shade")) else walk
            fam_bytes[fam] = fam_bytes.get(fam, 0.0) + (fetch_factor * fetch[k] + write_factor * write.get(k, 0.0)) * 1024 * nf[k] / steps_pmc
        by_kernel = {fam: {"hbm_bytes_per_step": round(v), "algorithmic_bytes_per_step": alg[fam], "ratio": round(v / alg[fam], 3)}
                     for fam, v in fam_bytes.items() if alg.get(fam)}
except (OSError, KeyError, ValueError):
    pass
allrec[config] = {
    "kernel": kernel, "calibration_file": calibration, "traffic_over_algorithmic_by_kernel": by_kernel, "hbm_bytes_per_launch": round(num / max(sum(nf.values()), 1)), "kernel_source_sha16": kernel_source_hash(),
    "valu_issue_busy": valu_busy, "lane_active": lane_active,
    "how": f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py --config {config} --steps 2 "
           f"--warmup 0 --no-cpu-baseline` (tools/profile_bench.sh {tag}); bytes = ({fetch_factor:g}*FETCH_SIZE + {write_factor:g}*WRITE_SIZE)*1024 "
           f"per dispatch, averaged over the {kernel} dispatches; FETCH factor: tools/pmc_calibrate.sh on the kernels' own read "
           f"patterns ({calibration})",
    "fetch_factor": fetch_factor, "write_factor": write_factor,
    "fetch_kib_per_dispatch": fetch, "write_kib_per_dispatch": write, "dispatches": nf, "tag": tag,
}
json.dump(allrec, open(path, "w"), indent=1)
print(json.dumps(allrec[config], indent=1))
