"""copy the judged summaries of gpurun_out/prof_<tag>/ (tools/profile_bench.sh) into profiles/ and refresh pmc_traffic.json"""
import csv, glob, json, os, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
shutil.copy(os.path.join(src, "summary.txt"), os.path.join(dst, f"{tag}_rocprofv3_summary.txt"))
shutil.copy(os.path.join(src, "summary.json"), os.path.join(dst, f"{tag}_rocprofv3_summary.json"))
ks = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
if ks:
    shutil.copy(max(ks, key=os.path.getmtime), os.path.join(dst, f"{tag}_kernel_stats.csv"))
b = os.path.join(root, "gpurun_out", "bench_n1.json")
if os.path.exists(b):
    shutil.copy(b, os.path.join(dst, f"{tag}_bench_n1.json"))


def per_dispatch(sub, ctr):
    f = max(glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)  # newest run
    tot, n = {}, {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != ctr or "k_bounce" not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        tot[k] = tot.get(k, 0.0) + float(r["Counter_Value"])
        n[k] = n.get(k, 0) + 1
    return {k: tot[k] / n[k] for k in tot}, n


fetch, nf = per_dispatch("pmc_fetch", "FETCH_SIZE")
write, nw = per_dispatch("pmc_write", "WRITE_SIZE")
num = sum((2 * fetch[k] + write[k]) * 1024 * nf[k] for k in fetch)
out = {
    "k_bounce_hbm_bytes_per_launch": round(num / sum(nf.values())),
    "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py --steps 2 --warmup 0 "
           "--no-cpu-baseline` (tools/profile_bench.sh); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per dispatch (gfx950 FETCH_SIZE "
           "counts half of coalesced streaming reads, MI355X_MICROARCH.md 'HBM'), averaged over the k_bounce<true,0> and "
           "k_bounce<false,0> dispatches",
    "fetch_kib_per_dispatch": fetch, "write_kib_per_dispatch": write, "round": int(tag[1:3]) if tag[1:3].isdigit() else None,
}
json.dump(out, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
